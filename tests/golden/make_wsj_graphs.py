#!/usr/bin/env python3
"""Converts the reference's graph fixtures misc/benchmark/{den,num}_fsm_wsj.txt
(OpenFst text written by misc/benchmark/generatefsm.jl:42-57; MIT licence) into
compact .npz arc lists -- read by the PRODUCT reader (FSM.from_openfst_text), checked against the oracle's parser
and the product writer -- and writes oracle outputs on them.

Needs /root/reference (build container only); the .npz files are committed.
Run: python tests/golden/make_wsj_graphs.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402

REF = "/root/reference/misc/benchmark"


def main():
    o, oc = ge.load_oracle()
    import graphs

    mm = ge.load_package()
    import importlib

    wl = importlib.import_module(mm.__name__ + ".workloads")
    for name in ("den_fsm_wsj", "num_fsm_wsj"):
        text = open(os.path.join(REF, name + ".txt")).read()
        # through the PRODUCT reader (markovmodels.jl_amd/fsm.py: FSM.from_openfst_text); the oracle's parser must agree
        fsm, s2p, P = mm.FSM.from_openfst_text(text)
        ii, iw, src, dst, w, fi, fw = fsm.arc_lists()
        S = fsm.S1 - 1
        So, init, arcs, final, s2p_o, P_o = o.parse_openfst_text(text)
        assert (S, P) == (So, P_o) and np.array_equal(s2p, s2p_o)
        assert np.array_equal(ii, [s for s, _ in init]) and np.array_equal(iw, np.array([x for _, x in init], dtype=np.float32))
        assert np.array_equal(src, [a[0][0] for a in arcs]) and np.array_equal(dst, [a[0][1] for a in arcs])
        assert np.array_equal(w, np.array([a[1] for a in arcs], dtype=np.float32))
        assert np.array_equal(fi, [s for s, _ in final]) and np.array_equal(fw, np.array([x for _, x in final], dtype=np.float32))
        assert fsm.to_openfst_text(s2p) == text  # the product writer gives the reference's file back, byte for byte
        np.savez_compressed(
            os.path.join(HERE, name + ".npz"), S=S, P=P,
            init_idx=ii.astype(np.int32), init_w=iw.astype(np.float32),
            src=src.astype(np.int32), dst=dst.astype(np.int32), w=w.astype(np.float32),
            final_idx=fi.astype(np.int32), final_w=fw.astype(np.float32),
            state2pdf=s2p.astype(np.int16),
        )
        g = wl.load_npz_graph(os.path.join(HERE, name + ".npz"))
        print(name, "S", g.S, "arcs", g.src.size, "init", g.init_idx.size, "final", g.final_idx.size, "P", g.P)
        # oracle outputs (float64 restatement) on seeded inputs
        rng = np.random.default_rng(2024)
        # the numerator graph needs >= 166 frames to reach a final state; one utterance
        # is deliberately too short (no accepting path: the reference's 0/0 -> NaN case)
        B, N = (3, 40) if name.startswith("den") else (3, 200)
        V = rng.standard_normal((B, N, g.P)).astype(np.float32)
        lens = np.array([N, N - 11, N - 23] if name.startswith("den") else [N, 180, 150], dtype=np.int32)
        gam, ttl = oc.batch_shared(graphs.to_oracle(o, g), g.state2pdf, g.P, V, lens, dtype=np.float64)
        path, score, _ = oc.viterbi(graphs.to_oracle(o, g, "tropical", np.float32), g.state2pdf, g.P, V[0], N,
                                    dtype=np.float32)
        np.savez_compressed(os.path.join(HERE, name + "_oracle.npz"), V=V, lens=lens, gamma=gam.astype(np.float32),
                            ttl=ttl, path=path, score=score)


if __name__ == "__main__":
    main()
