#!/usr/bin/env python
"""Per-layer table (time, TF/s, GB/s) of one trunk pass from a rocprofv3 kernel-trace sqlite db.
usage: tools/layer_profile.py results.db images_per_chunk"""
import sqlite3
import sys

db, n = sys.argv[1], int(sys.argv[2])
FUSED_DS = True   # (the downsample branch is always folded into conv3 now)
c = sqlite3.connect(db)
rows = list(c.execute("select name, start, end, duration, grid_x from kernels order by start"))
idx = [i for i, r in enumerate(rows) if "stem" in r[0]]
i0 = idx[-2]
layers, planes = (3, 4, 6, 3), (64, 128, 256, 512)
shapes, inpl, H = [], 64, 56
for li, (pl, nb) in enumerate(zip(planes, layers)):
    for bi in range(nb):
        st = 2 if (bi == 0 and li > 0) else 1
        Ho = H // st
        shapes.append(("l%d.%d.c1" % (li + 1, bi), n * H * H, pl, inpl, n * H * H * inpl, 0))
        shapes.append(("l%d.%d.c2" % (li + 1, bi), n * Ho * Ho, pl, pl * 9, n * H * H * pl, 0))
        if bi == 0 and not FUSED_DS:
            shapes.append(("l%d.%d.ds" % (li + 1, bi), n * Ho * Ho, pl * 4, inpl, n * H * H * inpl // (st * st), 0))
        if bi == 0 and FUSED_DS:   # conv3 + downsample as one GEMM over K = planes + inplanes
            shapes.append(("l%d.%d.c3d" % (li + 1, bi), n * Ho * Ho, pl * 4, pl + inpl,
                           n * Ho * Ho * pl + n * H * H * inpl // (st * st), 0))
        else:
            shapes.append(("l%d.%d.c3" % (li + 1, bi), n * Ho * Ho, pl * 4, pl, n * Ho * Ho * pl, 1))
        inpl, H = pl * 4, Ho
k = i0 + 1
while "conv_" not in rows[k][0] and "bneck" not in rows[k][0] and "blk_img" not in rows[k][0]:
    print("%-40s %8.1f us" % (rows[k][0][:40], rows[k][3] / 1e3))
    k += 1
print("%-40s %8.1f us" % (rows[i0][0][:40], rows[i0][3] / 1e3))
tot = 0
bylayer = {}
if "bneck" in rows[k][0]:        # layer1 as three fused bottleneck kernels: fold its 9 shape rows into 3
    fused, rest = [], []
    for sh in shapes:
        (fused if sh[0].startswith("l1.") else rest).append(sh)
    for bi in range(3):
        blk = [sh for sh in fused if sh[0].startswith("l1.%d." % bi)]
        fl = sum(2.0 * M * N * K for (_, M, N, K, _, _) in blk)
        M = blk[0][1]
        by = (M * blk[0][3] + M * 256) * 2          # block input + output, once each
        r = rows[k]
        k += 1
        dur = r[3] / 1e3
        tot += dur
        bylayer["l1"] = bylayer.get("l1", 0) + dur
        tail = "<false, true>" in r[0]              # tail variant: + conv1 of layer2.0; the block output at the even pixels only
        if tail:
            _, M1, N1, K1, _, _ = rest[0]
            assert rest[0][0] == "l2.0.c1"
            fl += 2.0 * M1 * N1 * K1
            by = (M * blk[0][3] + M * 256 // 4 + M1 * N1) * 2
            rest = rest[1:]
        print("l1.%d.fused M=%7d (conv1+conv2+conv3%s%s) %-22s grid=%6d %7.1fus %7.1f TF/s %7.1f MB %6.0f GB/s" % (
            bi, M, "+ds" if bi == 0 else "+id", "+l2.0.c1" if tail else "", r[0].split("::")[-1].split("(")[0], r[4] // 512, dur,
            fl / dur / 1e6, by / 1e6, by / dur / 1e3))
    shapes = rest
import re
skip = set()
for si, (nm, M, N, K, inel, res) in enumerate(shapes):
    if si in skip:
        continue
    r = rows[k]
    k += 1
    if "blk_img_kernel" in r[0]:
        # image-resident identity block: conv1 + conv2 + conv3 + identity of THIS block in one launch (x read twice, out once)
        blk = [shapes[j] for j in (si, si + 1, si + 2)]
        assert nm.endswith(".c1") and blk[2][0].endswith(".c3"), (nm, blk)
        skip.update((si + 1, si + 2))
        fl, dur = sum(2.0 * m_ * n_ * k_ for (_, m_, n_, k_, _, _) in blk), r[3] / 1e3
        by = (2 * M * K + M * K) * 2
        tot += dur
        bylayer[nm[:2]] = bylayer.get(nm[:2], 0) + dur
        print("%-16s M=%7d (conv1+conv2+conv3+id)   %-22s grid=%6d %7.1fus %7.1f TF/s %7.1f MB %6.0f GB/s" % (
            nm[:-3] + ".img", M, "blk_img_kernel", r[4] // 256, dur, fl / dur / 1e6, by / 1e6, by / dur / 1e3))
        continue
    assert "conv_" in r[0], r[0]
    fl, dur = 2.0 * M * N * K, r[3] / 1e3
    by = (inel + M * N * (1 + res)) * 2
    if "conv_pair_kernel" in r[0]:
        # conv3 (+ identity | + folded downsample) of this block and -- N1 > 0 -- conv1 of the next block in one launch:
        # template arguments <P, P2, C3, N1, ...>; the merged row carries both layers' FLOPs and the fused kernel's bytes
        targs = [int(v) for v in re.findall(r"-?\d+", r[0].split("<")[1])[:4]]
        n1 = targs[3]
        cfg = "pair<%d,%d,%d,%d>" % tuple(targs)
        if n1 > 0:
            nx = next(j for j in range(si + 1, len(shapes)) if shapes[j][0].endswith(".c1"))
            _, M1, N1s, K1, _, _ = shapes[nx]
            assert N1s == n1 and K1 == N and M1 == M, (shapes[nx], targs)
            skip.add(nx)
            fl += 2.0 * M1 * N1s * K1
            by += M1 * N1s * 2                              # + the next block's conv1 output; its input never leaves the chip
            nm = nm + "+" + shapes[nx][0].split(".", 2)[0][1:] + "." + shapes[nx][0].split(".")[1] + ".c1"
    else:
        cfg = r[0].split("<")[1].split(">")[0].replace("unsigned short", "bf16") if "<" in r[0] else r[0].split("::")[-1].split("(")[0]
        if "conv_lean_kernel" in r[0]:
            cfg = "conv_lean_kernel" + ("<pool>" if "<true>" in r[0] else "")
    tot += dur
    bylayer[nm[:2]] = bylayer.get(nm[:2], 0) + dur
    print("%-16s M=%7d N=%4d K=%4d %-22s grid=%6d %7.1fus %7.1f TF/s %7.1f MB %6.0f GB/s" % (
        nm, M, N, K, cfg, r[4] // 256, dur, fl / dur / 1e6, by / 1e6, by / dur / 1e3))
print("total conv us", tot, bylayer)
