"""Random shortint circuits (lin / pbs DAGs) with their clear-text semantics; test infrastructure."""
import numpy as np


def build_random_circuit(plan, rng, n_inputs=6, n_ops=40, M=4, T=16):
    """Returns (input node ids, output node ids, clear evaluator(values) -> outputs)."""
    nodes, prog = [], []
    for _ in range(n_inputs):
        nodes.append(plan.input(M - 1))
        prog.append(("in",))
    degree = [M - 1] * n_inputs
    luts = {}

    def lut(table):
        key = tuple(table)
        if key not in luts:
            luts[key] = plan.lut(lambda x, t=table: t[x])
        return luts[key]

    for _ in range(n_ops):
        kind = rng.choice(["pbs", "pack", "sum"])
        if kind == "pbs":
            src = int(rng.integers(0, len(nodes)))
            table = [int(v) for v in rng.integers(0, M, size=T)]
            nodes.append(plan.pbs(nodes[src], lut(table)))
            prog.append(("pbs", src, table))
            degree.append(M - 1)
        elif kind == "pack":   # bivariate: a*M + b -> table
            cands = [i for i, d in enumerate(degree) if d <= M - 1]
            a, b = (int(v) for v in rng.choice(cands, size=2))
            table = [int(v) for v in rng.integers(0, M, size=T)]
            packed = plan.lin([(nodes[a], M), (nodes[b], 1)])
            nodes.append(plan.pbs(packed, lut(table)))
            prog.append(("pack", a, b, table))
            degree.append(M - 1)
        else:                  # sum of up to 3 bounded nodes + constant, then identity-mod table
            cands = [i for i, d in enumerate(degree) if d <= M - 1]
            picks = [int(v) for v in rng.choice(cands, size=int(rng.integers(1, 4)))]
            cst = int(rng.integers(0, 3))
            table = [int(v) for v in rng.integers(0, M, size=T)]
            summed = plan.lin([(nodes[i], 1) for i in picks], cst)
            nodes.append(plan.pbs(summed, lut(table)))
            prog.append(("sum", picks, cst, table))
            degree.append(M - 1)
    outs = [int(v) for v in rng.choice(np.arange(n_inputs, len(nodes)), size=5, replace=False)]
    for o in outs:
        plan.output(nodes[o])

    def evaluate(values):
        vals = list(values)
        for step in prog[n_inputs:]:
            if step[0] == "pbs":
                vals.append(step[2][vals[step[1]]])
            elif step[0] == "pack":
                vals.append(step[3][vals[step[1]] * M + vals[step[2]]])
            else:
                vals.append(step[3][sum(vals[i] for i in step[1]) + step[2]])
        return [vals[o] for o in outs]

    return evaluate
