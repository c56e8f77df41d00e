#!/usr/bin/env python3
"""size the polyphase form of configs[4] (N = 2048) from an ACTUAL COMPILE of the candidate kernel
(tools/sizing/p2048_polyphase_candidate.hip) beside the split form that runs today, with the flags of build.sh:
registers, spills, LDS, occupancy, and the static VALU mix of one blind-rotate step (tools/isa_mix.py's path analysis),
priced with profiles/valu_issue_costs.json.

    python tools/sizing/p2048_polyphase.py > profiles/r06_p2048_polyphase_sizing.txt
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_mix  # noqa: E402

FLAGS = ["-O3", "-std=c++17", "-ffp-contract=off", "-mllvm", "-amdgpu-sched-strategy=max-ilp", "--offload-arch=gfx950", "--cuda-device-only", "-S"]


def compile_asm(src):
    out = os.path.join(tempfile.gettempdir(), os.path.basename(src) + ".s")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + [src, "-o", out], stderr=subprocess.DEVNULL)
    return open(out).read()


def meta(asm, needle):
    blk = asm[asm.index(".amdhsa_kernel " + next(m.group(1) for m in re.finditer(r"\.amdhsa_kernel (\S+)", asm) if needle in m.group(1))):]
    res = {}
    tail = asm[asm.index("; Kernel info:", asm.index(needle + "") if False else 0):] if False else asm
    # the per-kernel summary comments follow the kernel's code
    start = next(m.start() for m in re.finditer(r"^(_Z\w+):", asm, re.M) if needle in m.group(1))
    seg = asm[start:]
    for key in ("NumVgprs", "ScratchSize", "Occupancy", "LDSByteSize", "NumSgprs"):
        m = re.search(r"; %s: (\d+)" % key, seg)
        res[key] = int(m.group(1)) if m else None
    return res


def priced(v, costs, col):
    return sum(v[c] * costs[c][col] for c in ("mul", "three_operand", "two_operand"))


def main():
    costs = json.load(open(os.path.join(ROOT, "profiles", "valu_issue_costs.json")))["classes"]
    cand_asm = compile_asm(os.path.join(ROOT, "tools", "sizing", "p2048_polyphase_candidate.hip"))
    prod_asm = compile_asm(os.path.join(ROOT, "peba1_amd", "csrc", "kernels.hip"))
    isa_mix.KERNELS["blind_rotate_polyphase_candidate"] = ("blind_rotate_polyphase_candidate", "polyphase form (candidate, N = 2048)",
                                                           [("wave", lambda l: l, 2)])
    cand = isa_mix.analyse(cand_asm.split("\n"), "blind_rotate_polyphase_candidate", 3)
    split = isa_mix.analyse(prod_asm.split("\n"), "blind_rotate_split_kernelILi11ELi2E", 3)
    rows = []
    for name, k, asm, needle in (("split form (runs today)", split, prod_asm, "blind_rotate_split_kernelILi11ELi2E"),
                                 ("polyphase candidate", cand, cand_asm, "blind_rotate_polyphase_candidate")):
        v = k["roles"][0]["variants"][0]
        m = meta(asm, needle)
        rows.append((name, v, m, 2 * priced(v, costs, "two_waves_per_simd"), 2 * priced(v, costs, "eight_waves_per_simd")))
    print("# configs[4] (N = 2048, l = 3, Bg = 2^6): the polyphase form sized from a compile of the candidate kernel")
    print("# tools/sizing/p2048_polyphase.py; flags of peba1_amd/csrc/build.sh; static mix of ONE blind-rotate step per wave (tools/isa_mix.py)")
    print()
    print(f"{'':28s} {'VGPRs':>6s} {'scratch':>8s} {'LDS B':>8s} {'waves/SIMD':>10s} | {'VALU':>5s} {'mul':>5s} {'3-op':>5s} {'2-op':>5s} {'LDS':>4s} {'vmem':>5s} {'bar':>3s} | "
          f"{'issue cycles / SIMD-step':>25s} {'at chip-peak rates':>19s}")
    for name, v, m, cyc, peak in rows:
        print(f"{name:28s} {m['NumVgprs']:6d} {m['ScratchSize']:8d} {m['LDSByteSize']:8d} {m['Occupancy']:10d} | {v['valu']:5d} {v['mul']:5d} {v['three_operand']:5d} "
              f"{v['two_operand']:5d} {v['lds']:4d} {v['vmem']:5d} {v['barriers']:3d} | {cyc:25.0f} {peak:19.0f}")
    (_, vs, _, cs, ps), (_, vc, _, cc, pc) = rows
    print()
    print(f"difference (candidate - split): VALU {vc['valu'] - vs['valu']:+d}, mul {vc['mul'] - vs['mul']:+d}, three-operand "
          f"{vc['three_operand'] - vs['three_operand']:+d}, two-operand {vc['two_operand'] - vs['two_operand']:+d}, LDS instructions {vc['lds'] - vs['lds']:+d}, "
          f"vector-memory instructions {vc['vmem'] - vs['vmem']:+d}")
    print(f"issue-cost model (two waves per SIMD): {cc:.0f} against {cs:.0f} cycles per SIMD and step = {cc / cs - 1:+.1%}")
    return rows


if __name__ == "__main__":
    main()
