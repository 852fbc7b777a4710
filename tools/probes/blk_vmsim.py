# issue-order simulation of conv3's vector-memory queue (loads and stores retire in order): for each counted wait, the number of
# operations younger than the one it must cover
KI = 2
ops = []     # (tag)
waits = {}
def issue(tag, n=1):
    for _ in range(n): ops.append(tag)
def younger(tag_pred):
    # index of the last op matching pred
    idx = max(i for i, t in enumerate(ops) if tag_pred(t))
    return len(ops) - 1 - idx
# conv2 tail: refills at the ends of its last 4 steps (global step numbering: conv3 step S = c*8+ks; conv2 last steps = -4..-1)
for s in range(-4, 0): issue(("ring", s + 4), 4)      # refill issued at end of step s brings the piece for step s+4
for c in range(3):
    for ks in range(8):
        S = c * 8 + ks
        n = younger(lambda t: t == ("ring", S))
        waits[("ring", c, ks)] = n
        if ks == KI:
            issue(("id0", c), 14); issue(("bn", c), 8)
        issue(("ring", S + 4), 4)
    waits[("E0", c)] = younger(lambda t: t == ("bn", c))
    for g in range(14):
        issue(("st0", c, g)); issue(("id1", c, g))
    for g in range(14):
        waits[("Q1", c, g)] = younger(lambda t, g=g, c=c: t == ("id1", c, g))
        issue(("st1", c, g))
for k in sorted(waits, key=str): print(k, waits[k])
