"""Expected values of the reference's `mul!` test (test/test_linalg.jl:88-108) from the DEFINITION of the three semirings
(Semirings.jl 0.5, un-vendored: Log x(+)y = log(e^x + e^y), x(*)y = x + y; Tropical (+) = max, (*) = +; Prob ordinary + and *),
in exact-ish arithmetic (math.fsum / math.log in float64) -- independent of oracle/ and of the product.  The reference's test
asserts GPU mul! == generic CPU mul! on these inputs; the generic CPU mul! is the definition evaluated here.

    sm = sparse([1, 2, 2, 3, 4], [3, 1, 2, 1, 3], K[1, 2, 3, 4, 5], 4, 3)
    dm = reshape(Array{K}(collect(1:12)), 3, 4);  dv = Array{K}(collect(1:3))
    mul!(similar(dm, 4, 4), sm, dm);  mul!(similar(dv, 4), sm, dv)

Writes the "expected" block into known_answers.json (mul_known_answer)."""
import json
import math
import os

HERE = os.path.dirname(os.path.abspath(__file__))
I, J, V = [1, 2, 2, 3, 4], [3, 1, 2, 1, 3], [1.0, 2.0, 3.0, 4.0, 5.0]
dv = [1.0, 2.0, 3.0]
dm = [[float(1 + r + 3 * c) for c in range(4)] for r in range(3)]  # reshape(1:12, 3, 4): column-major


def row_terms(r, vec):
    return [(v, vec[j - 1]) for i, j, v in zip(I, J, V) if i == r + 1]


def reduce_(K, terms):
    if K == "prob":
        return math.fsum(a * b for a, b in terms) if terms else 0.0
    xs = [a + b for a, b in terms]
    if not xs:
        return -math.inf
    if K == "tropical":
        return max(xs)
    m = max(xs)
    return m + math.log(math.fsum(math.exp(x - m) for x in xs))


def main():
    exp = {}
    for K in ("log", "tropical", "prob"):
        spmv = [reduce_(K, row_terms(r, dv)) for r in range(4)]
        spmm = [[reduce_(K, row_terms(r, [dm[k][c] for k in range(3)])) for c in range(4)] for r in range(4)]
        exp[K] = {"spmv": spmv, "spmm_rows": spmm}
    path = os.path.join(HERE, "known_answers.json")
    ka = json.load(open(path))
    ka["mul_known_answer"]["expected"] = exp
    ka["mul_known_answer"]["expected_source"] = "tests/golden/make_mul_fixture.py: the semirings' definitions in float64 (zero(K) for the empty sums: none here)"
    with open(path, "w") as f:
        json.dump(ka, f, indent=1)
        f.write("\n")
    print(json.dumps(exp, indent=1))


if __name__ == "__main__":
    main()
