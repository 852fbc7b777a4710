"""LDS fragment layouts of the HIP kernels against ds_read_b128's lane groups (MI355X_MICROARCH.md, LDS section): the rule the kernels'
layouts are built on, checked on the CPU (tools/probes/lds_groups.py restates it).  Three layouts of rounds 2-5 were designed for
16 CONTIGUOUS lanes per LDS cycle and were 2-way conflicted in every real group; this pins the ones in the tree."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "probes"))
import lds_groups as G  # noqa: E402


def test_the_lane_groups_partition_a_wave():
    lanes = sorted(l for g in G.GROUPS for l in g)
    assert lanes == list(range(64)) and all(len(g) == 16 for g in G.GROUPS)


def test_stem_weight_rows_are_conflict_free():
    """stem.hip: lane (row lr, K group g) reads 16 bytes at lr * 2 * AP_STEM_WLD + 16 g (ap_common.h: AP_STEM_WLD = 240)."""
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "airpose_amd", "csrc", "ap_common.h")).read()
    wld = int(src.split("#define AP_STEM_WLD")[1].split()[0])
    assert G.extra_cycles(lambda l: (l & 15) * 2 * wld + (l >> 4) * 16) == 0
    assert G.extra_cycles(lambda l: (l & 15) * 464 + (l >> 4) * 16) == 4          # rounds 2-5: every group 2-way


def test_image_slot_rotation_is_conflict_free_under_every_tap_shift():
    """conv_img3.hip / block_img.hip: chunk c of slot u at position (c + 2 u) mod 16; the XOR form of rounds 5-6 was not."""
    assert G.image_slots(lambda m, c: (c + 2 * m) & 15) == 0
    assert G.image_slots(lambda m, c: c ^ (m & 15)) == 4
