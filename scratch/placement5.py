"""placement3 cut down for runs under rocprofv3 --pmc: fixed S and Y, eight fresh allocations of g, four two-loops each."""
import os, sys
sys.argv = [sys.argv[0]]
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "placement3.py")).read().split("S, Y = alloc(8 * m * n), alloc(8 * m * n)")[0]
exec(src)
S, Y = alloc(8 * m * n), alloc(8 * m * n)
fill(S, Y)
held = []
for rep in range(8):
    g = alloc(8 * n)
    print(json.dumps({"rep": rep, "g": hex(g), **measure(S, Y, g, reps=4)}), flush=True)
    held.append(g)
    held.append(alloc((rep + 1) * 77_594_624))
