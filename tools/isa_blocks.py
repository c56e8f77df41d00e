#!/usr/bin/env python3
"""Static instruction histogram of one kernel of a gfx950 assembly listing, per basic block:
   hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 --cuda-device-only -S kernels.hip -o k.s
   tools/isa_blocks.py k.s blind_rotate4_kernelILi10
Prints, for every basic block, its size, VALU / multiplier-class / LDS / global counts and where it
branches, and the multiplier-class mnemonics of the whole kernel.  Used to count the instructions of
one blind-rotate step (the loop blocks, weighted by hand with their trip counts)."""
import collections
import re
import sys

MULS = ("v_mad_i64_i32", "v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_i32", "v_mul_hi_u32", "v_mul_u32_u24",
        "v_mad_u32_u24", "v_mad_i32_i24", "v_mul_i32_i24")


def main():
    txt = open(sys.argv[1]).read().split("\n")
    needle = sys.argv[2]
    start = next(i for i, l in enumerate(txt) if needle in l and re.match(r"^_Z\w+:", l))
    end = next(i for i in range(start, len(txt)) if "s_endpgm" in txt[i])
    name, blocks, cur = "entry", [], []
    for l in txt[start + 1:end + 1]:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append((name, cur))
            name, cur = m.group(1), []
            continue
        s = l.strip()
        if s and not s.startswith((";", ".", "//")):
            cur.append(s)
    blocks.append((name, cur))
    total = collections.Counter()
    for name, ins in blocks:
        c = collections.Counter(i.split()[0] for i in ins)
        total.update(c)
        valu = sum(v for k, v in c.items() if k.startswith("v_"))
        mul = sum(c[k] for k in MULS)
        br = [i for i in ins if i.startswith(("s_cbranch", "s_branch"))]
        print(f"{name:12s} insts {len(ins):5d}  valu {valu:5d}  mul-class {mul:4d}  ds {sum(v for k, v in c.items() if k.startswith('ds_')):4d}"
              f"  global {sum(v for k, v in c.items() if k.startswith('global_')):3d}  salu {sum(v for k, v in c.items() if k.startswith('s_')):4d}"
              f"  -> {' | '.join(b.split()[-1] for b in br)}")
    print("whole kernel:", {k: total[k] for k in MULS if total[k]}, "valu", sum(v for k, v in total.items() if k.startswith("v_")))
    if len(sys.argv) > 3:
        want = sys.argv[3].split(",")
        for name, ins in blocks:
            if name in want:
                c = collections.Counter(i.split()[0] for i in ins)
                print(name, dict(c.most_common(40)))


if __name__ == "__main__":
    main()
