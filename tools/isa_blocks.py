#!/usr/bin/env python3
"""Basic-block instruction census of one kernel of a gfx950 assembly listing (hipcc -S --cuda-device-only):
   python tools/isa_blocks.py <file.s> <mangled kernel name substring> [-v]
Per block: label, lines, MFMA / other vector / DPP / scalar / LDS / vector-memory / branch / waitcnt counts.  Used for the headline kernel's
instruction-group table ("the floor")."""
import re
import sys
from collections import Counter, OrderedDict


def classify(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_cbranch") or op == "s_branch" or op.startswith("s_setpc") or op.startswith("s_swappc"): return "branch"
    if op.startswith("s_waitcnt") or op == "s_nop" or op.startswith("s_sleep") or op.startswith("s_setprio") or op.startswith("s_barrier"): return "wait"
    if op.startswith("s_load") or op.startswith("s_buffer_load") or op.startswith("s_memtime") or op.startswith("s_store"): return "smem"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"): return "vmem"
    return "other"


def kernel_lines(path, name):
    out = []; on = False
    for line in open(path):
        if not on:
            if line.startswith("_Z") and name in line.split(":")[0] and line.rstrip().split(";")[0].strip().endswith(":"):
                on = True
            continue
        if line.startswith("\t.section") or line.startswith(".Lfunc_end") or line.startswith("\t.end_amdhsa_kernel"):
            break
        out.append(line.rstrip("\n"))
    return out


def blocks(lines):
    bl = OrderedDict(); cur = "entry"; bl[cur] = []
    for l in lines:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            cur = m.group(1); bl[cur] = []
            continue
        t = l.strip()
        if not t or t.startswith(";") or t.startswith("."):
            continue
        bl[cur].append(t.split(";")[0].strip())
    return bl


if __name__ == "__main__":
    path, name = sys.argv[1], sys.argv[2]
    verbose = "-v" in sys.argv
    bl = blocks(kernel_lines(path, name))
    tot = Counter()
    print(f"{'block':>12} {'n':>5} {'mfma':>5} {'valu':>5} {'dpp':>4} {'salu':>5} {'lds':>4} {'vmem':>5} {'br':>3} {'wait':>4}  targets")
    for lab, ins in bl.items():
        c = Counter(classify(i.split()[0]) for i in ins)
        dpp = sum(1 for i in ins if "dpp" in i or "permlane" in i.split()[0] or "readlane" in i.split()[0] or "writelane" in i.split()[0])
        tg = [i.split()[-1] for i in ins if i.startswith("s_cbranch") or i.startswith("s_branch")]
        tot.update(c); tot["dpp"] += dpp
        print(f"{lab:>12} {len(ins):5d} {c['mfma']:5d} {c['valu']:5d} {dpp:4d} {c['salu']:5d} {c['lds']:4d} {c['vmem']:5d} {c['branch']:3d} {c['wait']:4d}  {' '.join(tg)}")
        if verbose:
            oc = Counter(i.split()[0] for i in ins)
            print("             " + ", ".join(f"{k} {v}" for k, v in oc.most_common(14)))
    print("total", dict(tot))
