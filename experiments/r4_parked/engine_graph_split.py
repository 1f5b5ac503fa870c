"""PARKED: ForwardEngine._run_split (TDEED_GRAPH_SPLIT=1): the sub-batches of a plan as separate single-chain graphs joined by
events instead of a fork inside one graph.  Measured a tie with 4 hardware queues and a loss with 8 (DESIGN history)."""

    def _run_split(self, plan, st):
        """Several sub-batches as SEPARATE single-chain graphs, one per stream, joined by events in front of the tail's graph.
        A fork INSIDE one graph replays at ~3.7 us per TRIVIAL kernel node over both branches together
        (tools/bench_dispatch.py: 256 nodes on two branches 960 us per replay), two single-chain graphs on two streams at 0.9 us
        per node in aggregate.  With real kernels the dispatch hides behind the other stream's execution: measured a tie with
        four hardware queues and a loss with eight, so this path is opt-in (TDEED_GRAPH_SPLIT=1)."""
        streams = [st if (s_ is None or s_.cuda_stream == st.cuda_stream) else s_ for s_ in plan.streams]
        if plan.graph is None:
            _drain_dead_graphs()
            self._launch_all(plan, st)         # warm-up launch (module load, validates arguments)
            st.synchronize()
            gs = []
            for sb, s_ in zip(plan.subs, streams):
                s_.synchronize()
                with torch.cuda.stream(s_):
                    gs.append(self._capture(s_, lambda sb=sb: [x.fn() for x in sb.steps]))
            gt = self._capture(st, lambda: [x.fn() for x in plan.tail.steps]) if plan.tail is not None else None
            plan.graph = SimpleNamespace(subs=gs, tail=gt, fork=torch.cuda.Event(),
                                         joins=[torch.cuda.Event() for _ in streams])
        g = plan.graph
        g.fork.record(st)
        for i, s_ in enumerate(streams):
            if s_.cuda_stream != st.cuda_stream:
                s_.wait_event(g.fork)
            _lib.call("tdeed_graph_launch", g.subs[i], s_.cuda_stream)
            if s_.cuda_stream != st.cuda_stream:
                g.joins[i].record(s_)
        for i, s_ in enumerate(streams):
            if s_.cuda_stream != st.cuda_stream:
                st.wait_event(g.joins[i])
        if g.tail is not None:
            _lib.call("tdeed_graph_launch", g.tail, st.cuda_stream)

