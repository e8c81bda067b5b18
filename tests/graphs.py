"""Test helpers: GraphSpec (product workloads) -> oracle FSM."""
import numpy as np


def to_oracle(o, g, semiring="log", dtype=np.float64):
    """GraphSpec -> oracle FSM through the reference's arc-list constructor
    restatement (mm_oracle.make_fsm)."""
    K = o.SEMIRINGS[semiring]
    return o.make_fsm(
        K,
        list(zip(g.init_idx.tolist(), g.init_w.tolist())),
        [((int(i), int(j)), float(w)) for i, j, w in zip(g.src, g.dst, g.w)],
        list(zip(g.final_idx.tolist(), g.final_w.tolist())),
        list(range(g.S)),
        dtype,
    )
