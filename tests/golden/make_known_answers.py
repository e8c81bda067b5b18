#!/usr/bin/env python3
"""Writes tests/golden/known_answers.json: the known-answer vectors the
reference's own tests / demo notebook hold for the inference path, transcribed
by hand (the reference is Julia and cannot run in the build image).  Each entry
cites its source (paths relative to the MarkovModels.jl tree).

Run: python tests/golden/make_known_answers.py
"""
import json
import math
import os

HERE = os.path.dirname(os.path.abspath(__file__))

answers = {
    "demo_notebook_gamma": {
        "source": "examples/demo.ipynb cells 5-13: 3-state left-to-right HMM (self loop + forward arc, renormalised, "
                  "initial state 1, final state 3), lhs = zeros(3, 5); printed output of pdfposteriors",
        "fsm": "l2r3",
        "lhs": "zeros(3,5)",
        "gamma": [[1.0, 0.5, 0.166667, 0.0, 0.0], [0.0, 0.5, 0.666667, 0.5, 0.0], [0.0, 0.0, 0.166667, 0.5, 1.0]],
        "gamma_exact": [[1, 0.5, 1 / 6, 0, 0], [0, 0.5, 2 / 3, 0.5, 0], [0, 0, 1 / 6, 0.5, 1]],
        "atol": 1e-6,
        "ttl_derived": math.log(6 / 32),
    },
    "batch_varlen": {
        "source": "test/test_algorithms.jl:218-248 (disabled suite): the same FSM twice, lhs = ones(3, 7), "
                  "seqlengths [5, 7]; expectations: gamma_1[:, 1:5] = single-utterance result on 5 frames, "
                  "gamma_1[:, 6:7] == 0 exactly, gamma_2 = single-utterance result on 7 frames, each checked "
                  "against the dense logsumexp forward/backward of test/test_algorithms.jl:28-63",
        "fsm": "l2r3",
        "lhs": "ones(3,7)",
        "seqlengths": [5, 7],
    },
    "bestpath_chain": {
        "source": "test/test_algorithms.jl:262-284: 4-state chain a->b->c->d (tropical), lhs = ones(4, 4): "
                  "best path reads 'a b c d'",
        "path_1based": [1, 2, 3, 4],
    },
    "mul_known_answer": {
        "source": "test/test_linalg.jl:88-108: sm = sparse([1,2,2,3,4],[3,1,2,1,3],K[1,2,3,4,5],4,3), "
                  "dm = reshape(K.(1:12),3,4), dv = K.(1:3); GPU mul! must equal the generic CPU mul!",
        "I": [1, 2, 2, 3, 4], "J": [3, 1, 2, 1, 3], "V": [1, 2, 3, 4, 5], "shape": [4, 3],
        "dv": [1, 2, 3],
        "dm_colmajor": list(range(1, 13)),
    },
    "logaddexp": {
        "source": "test/test_semirings.jl:4-6",
        "cases": [[2.0, 3.0, math.log(math.exp(2.0) + math.exp(3.0))],
                  [10002.0, 10003.0, 10000 + math.log(math.exp(2.0) + math.exp(3.0))]],
    },
}

with open(os.path.join(HERE, "known_answers.json"), "w") as f:
    json.dump(answers, f, indent=1)
print("wrote known_answers.json")
