"""Synthetic graphs for the BASELINE.json configurations and
readers for the packaged real graphs.  Pure NumPy; produces plain arc lists
(``GraphSpec``) that both the product (``to_fsm``) and the test oracle consume.
Weights are natural-log probabilities, rows normalised (sum over out-arcs +
final weight = 1) like the reference's ``renorm`` (src/fsmops.jl:71-79).
"""
from __future__ import annotations

import os
from dataclasses import dataclass

import numpy as np


@dataclass
class GraphSpec:
    name: str
    S: int  # real states (the FSM adds the phony final state)
    init_idx: np.ndarray
    init_w: np.ndarray
    src: np.ndarray
    dst: np.ndarray
    w: np.ndarray
    final_idx: np.ndarray
    final_w: np.ndarray
    state2pdf: np.ndarray  # [S] 0-based
    P: int

    @property
    def n_arcs(self) -> int:
        """stored entries of T_hat: arcs + final arcs + the final self loop"""
        return int(self.src.size + self.final_idx.size + 1)


def to_fsm(mm, g: GraphSpec, semiring: str = "log", dtype=np.float32):
    """GraphSpec -> product FSM (markovmodels.jl_amd.FSM)."""
    S = g.S
    I = np.concatenate([g.src, g.final_idx, [S]]).astype(np.int64)
    J = np.concatenate([g.dst, np.full(g.final_idx.size, S), [S]]).astype(np.int64)
    V = np.concatenate([g.w, g.final_w, [1.0 if semiring == "prob" else 0.0]]).astype(dtype)  # (the final self loop: one(K))
    f = mm.FSM.__new__(mm.FSM)
    f.semiring = semiring
    f.labels = list(range(S))
    f.dtype = np.dtype(dtype)
    f.alpha_idx = g.init_idx.astype(np.int64)
    f.alpha_val = g.init_w.astype(dtype)
    f._set_coo(I, J, V, S + 1)
    f._parts = None
    return f


def _normalise_rows(S, src, w_lin, final_idx, final_lin):
    tot = np.zeros(S)
    np.add.at(tot, src, w_lin)
    np.add.at(tot, final_idx, final_lin)
    return np.log(w_lin / tot[src]), np.log(final_lin / tot[final_idx])


def l2r_hmm(S: int = 3) -> GraphSpec:
    """Left-to-right HMM (self loop + forward arc), renormalised, initial state 0,
    final state S-1: the FSM of examples/demo.ipynb cell 5 and
    test/test_algorithms.jl:13-26.  One pdf per state."""
    src = np.array([s for s in range(S)] + [s for s in range(S - 1)])
    dst = np.array([s for s in range(S)] + [s + 1 for s in range(S - 1)])
    w, fw = _normalise_rows(S, src, np.ones(src.size), np.array([S - 1]), np.ones(1))
    return GraphSpec(f"l2r{S}", S, np.array([0]), np.array([0.0]), src, dst, w, np.array([S - 1]), fw,
                     np.arange(S, dtype=np.int32), S)


def random_fsm(S: int, P: int, mean_deg: float = 3.0, seed: int = 0, n_init: int = 2, p_final: float = 0.3) -> GraphSpec:
    """Random sparse graph with a guaranteed accepting chain 0 -> 1 -> ... -> S-1."""
    rng = np.random.default_rng(seed)
    deg = np.minimum(rng.poisson(mean_deg, S), S)
    src = np.concatenate([np.repeat(np.arange(S), deg), np.arange(S - 1)])
    dst = np.concatenate([rng.integers(0, S, int(deg.sum())), np.arange(1, S)])
    key = np.unique(src * S + dst)
    src, dst = key // S, key % S
    final_idx = np.unique(np.concatenate([np.flatnonzero(rng.random(S) < p_final), [S - 1]]))
    w, fw = _normalise_rows(S, src, rng.random(src.size) + 0.05, final_idx, rng.random(final_idx.size) + 0.05)
    init_idx = np.unique(np.concatenate([[0], rng.integers(0, S, n_init - 1)]))
    iw = rng.random(init_idx.size) + 0.1
    return GraphSpec(f"rand{S}", S, init_idx, np.log(iw / iw.sum()), src, dst, w, final_idx, fw,
                     rng.integers(0, P, S).astype(np.int32), P)


def dense_ergodic(S: int = 64, seed: int = 0) -> GraphSpec:
    """BASELINE config 2: dense ergodic HMM, T[i, :] = softmax(u_i) * S/(S+1), exit
    probability 1/(S+1) per state, uniform initial weights, identity state map."""
    rng = np.random.default_rng(seed)
    u = rng.standard_normal((S, S))
    T = np.exp(u - u.max(1, keepdims=True))
    T = T / T.sum(1, keepdims=True) * (S / (S + 1.0))
    src, dst = np.divmod(np.arange(S * S), S)
    return GraphSpec(f"ergodic{S}", S, np.arange(S), np.full(S, -np.log(S)), src, dst, np.log(T.ravel()),
                     np.arange(S), np.full(S, -np.log(S + 1.0)), np.arange(S, dtype=np.int32), S)


def lfmmi_denominator(S: int = 2000, P: int = 84, seed: int = 0, n_init: int = 38) -> GraphSpec:
    """BASELINE config 3: synthetic LF-MMI denominator graph mirroring the
    statistics of the reference's misc/benchmark/den_fsm_wsj.txt (3-gram
    phonotactic LM over 2-state phone HMMs): S/2 units of an entry state (no
    self loop) and an exit state (self loop); the exit state fans out to the
    entry states of ~32 successor units (clipped geometric, max 40) drawn with
    a heavy-tailed popularity, so out-degree has mean ~17 / max 41 and the
    in-degree is heavy tailed; 50 % of the states carry a self loop; n_init
    initial states; 30 % of the states (60 % of the exit states) are final;
    pdf = 2 * phone + position, P/2 phones."""
    rng = np.random.default_rng(seed)
    U = S // 2
    entry, exit_ = np.arange(U) * 2, np.arange(U) * 2 + 1
    phone = rng.integers(0, P // 2, U)
    s2p = np.empty(S, dtype=np.int32)
    s2p[entry], s2p[exit_] = 2 * phone, 2 * phone + 1
    deg = np.clip(rng.geometric(1.0 / 30.0, U) + 8, 1, 40)
    deg = np.minimum((deg * (32.0 / deg.mean())).round().astype(int), 40)
    pop = rng.lognormal(0.0, 0.8, U)
    pop /= pop.sum()
    succ = [rng.choice(U, size=int(d), replace=False, p=pop) for d in deg]
    src = np.concatenate([entry, exit_, np.repeat(exit_, deg)])
    dst = np.concatenate([exit_, exit_, entry[np.concatenate(succ)]])
    final_idx = exit_[rng.random(U) < 0.6]
    w, fw = _normalise_rows(S, src, rng.random(src.size) + 0.02, final_idx, 0.2 * rng.random(final_idx.size) + 0.01)
    init_idx = np.sort(entry[rng.choice(U, size=n_init, replace=False)])
    iw = rng.random(n_init) + 0.1
    return GraphSpec(f"lfmmi_den{S}", S, init_idx, np.log(iw / iw.sum()), src, dst, w, final_idx, fw, s2p, P)


def lexicon_fsm(S: int = 5000, P: int = 84, seed: int = 0, hubs: int = 4) -> GraphSpec:
    """BASELINE config 5: lexicon-like graph: `hubs` word-boundary states, word
    chains of 3-8 left-to-right states (self loop + forward arc) that start from
    a hub and return to one; mean out-degree ~2.2; initial and final = the hubs."""
    rng = np.random.default_rng(seed)
    src, dst = [], []
    s2p = np.empty(S, dtype=np.int32)
    s2p[:hubs] = np.arange(hubs) % P
    nxt = hubs
    for h in range(hubs):
        src.append(h), dst.append(h)
    wid = 0
    while nxt < S:
        L = int(min(rng.integers(3, 9), S - nxt))
        st = np.arange(nxt, nxt + L)
        s2p[st] = rng.integers(hubs, P, L)
        h_in, h_out = wid % hubs, int(rng.integers(0, hubs))
        src += [h_in] + st.tolist() + st[:-1].tolist() + [int(st[-1])]
        dst += [int(st[0])] + st.tolist() + st[1:].tolist() + [h_out]
        nxt += L
        wid += 1
    src, dst = np.asarray(src), np.asarray(dst)
    final_idx = np.arange(hubs)
    w, fw = _normalise_rows(S, src, rng.random(src.size) + 0.05, final_idx, np.full(hubs, 0.05))
    return GraphSpec(f"lexicon{S}", S, np.arange(hubs), np.full(hubs, -np.log(hubs)), src, dst, w, final_idx, fw, s2p, P)


def wide_row_fsm(S: int = 700, P: int = 11, seed: int = 0) -> GraphSpec:
    """A graph with one state of in-degree and one of out-degree > 256 (rows that
    need the long-row path of the kernels), for tests."""
    g = random_fsm(S, P, 2.0, seed)
    rng = np.random.default_rng(seed + 1)
    extra_in = np.setdiff1d(np.arange(S), [5])[: S - 50]
    src = np.concatenate([g.src, extra_in, np.full(S - 60, 7)])
    dst = np.concatenate([g.dst, np.full(extra_in.size, 5), np.arange(30, S - 30)])
    key = np.unique(src * S + dst)
    src, dst = key // S, key % S
    w, fw = _normalise_rows(S, src, rng.random(src.size) + 0.05, g.final_idx, np.exp(g.final_w))
    return GraphSpec(f"wide{S}", S, g.init_idx, g.init_w, src, dst, w, g.final_idx, fw, g.state2pdf, P)


def load_npz_graph(path: str) -> GraphSpec:
    """Graphs converted from the reference's OpenFst-text fixtures
    (tests/golden/make_wsj_graphs.py)."""
    z = np.load(path)
    return GraphSpec(os.path.basename(path).split(".")[0], int(z["S"]), z["init_idx"], z["init_w"], z["src"], z["dst"],
                     z["w"], z["final_idx"], z["final_w"], z["state2pdf"].astype(np.int32), int(z["P"]))


def sample_paths(g: GraphSpec, B: int, N: int, seed: int = 0, tail: int = 96) -> np.ndarray:
    """B accepting state paths of exactly N frames drawn from the graph's own distribution (initial weights, then the arcs'
    probabilities; the last `tail` frames restricted to the arcs from which a final state is still reachable in exactly the
    frames that are left, and the last state drawn by its final weight).  Returns states[B, N] (0-based).  What a TRAINED acoustic
    model's outputs are consistent with (examples/test_cuda.jl:124-143 feeds network outputs): `path_consistent_emissions`."""
    rng = np.random.default_rng(seed)
    S = g.S
    order = np.argsort(g.src, kind="stable")
    src, dst, p = g.src[order], g.dst[order], np.exp(g.w[order])
    deg = np.bincount(src, minlength=S)
    ptr = np.concatenate([[0], np.cumsum(deg)])
    D = int(deg.max())
    col = np.arange(src.size) - ptr[src]
    DST = np.zeros((S, D), dtype=np.int64)
    PRB = np.zeros((S, D))
    DST[src, col], PRB[src, col] = dst, p
    fin = np.zeros(S)
    fin[g.final_idx] = np.exp(g.final_w)
    # can[r][s]: from s, a final state can be reached in exactly r more arcs (r = 0: s is final)
    tail = int(min(tail, N - 1)) if N > 1 else 0
    can = [fin > 0]
    for _ in range(tail):
        can.append(((PRB > 0) & can[-1][DST]).any(axis=1))
    if N - 1 > tail and not can[tail][np.unique(dst)].all():
        raise ValueError("sample_paths: some state cannot reach a final state in `tail` arcs: raise `tail`")

    def draw(W):  # one column per row of W by its (unnormalised) weights
        c = np.cumsum(W, axis=1)
        if not (c[:, -1] > 0).all():
            raise ValueError("sample_paths: a path ran into a state from which no accepting path of the right length leaves")
        u = rng.random(W.shape[0]) * c[:, -1]
        return np.minimum((c < u[:, None]).sum(axis=1), W.shape[1] - 1)

    out = np.empty((B, N), dtype=np.int64)
    iw = np.exp(g.init_w) * (can[min(tail, N - 1)][g.init_idx] if N - 1 <= tail else 1.0)
    out[:, 0] = g.init_idx[draw(np.broadcast_to(iw, (B, iw.size)))]
    for n in range(1, N):
        left = N - 1 - n  # arcs still to come after this one
        s = out[:, n - 1]
        W = PRB[s]
        if left <= tail:
            W = W * (can[left][DST[s]] * (fin[DST[s]] if left == 0 else 1.0))
        out[:, n] = DST[s, draw(W)]
    return out


def path_consistent_emissions(g: GraphSpec, B: int, N: int, sigma: float, seed: int = 0, noise: float = 0.3) -> np.ndarray:
    """V[B, N, P] float32 = log-softmax(sigma * (onehot(pdf of the state of a sampled accepting path) + noise * N(0,1))): sharp AND
    consistent with the graph -- the forward and the backward mass of every frame meet on the path, as under a trained model
    (log-softmax of sigma * N(0,1) alone is sharp and inconsistent: a random arg-max sequence is not a path)."""
    rng = np.random.default_rng(seed + 7919)
    pdf = np.asarray(g.state2pdf)[sample_paths(g, B, N, seed)]
    x = noise * rng.standard_normal((B, N, g.P))
    np.put_along_axis(x, pdf[:, :, None], np.take_along_axis(x, pdf[:, :, None], 2) + 1.0, 2)
    x *= sigma
    x -= x.max(-1, keepdims=True)
    return (x - np.log(np.exp(x).sum(-1, keepdims=True))).astype(np.float32)
