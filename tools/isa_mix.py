#!/usr/bin/env python3
"""Static VALU instruction mix of ONE blind-rotate step, per wave role, from the gfx950 assembly of the kernels as built.

    hipcc <the flags of peba1_amd/csrc/build.sh> --cuda-device-only -S kernels.hip -o kernels.s
    tools/isa_mix.py kernels.s --l 3 [--json out.json]

`__graft_entry__.build()` runs exactly that and writes profiles/isa_mix_<kernels_sha16>.json; bench.py prices the mix with
the measured issue costs of profiles/valu_issue_costs.json and reports the result as `roofline.valu_issue` (VERDICT r4
item 2: the roofline that binds, reproducible from tracked files).

How a step is found (nothing is weighted by hand): the compiler annotates every basic block with the loop it belongs to
("in Loop: Header=BB20_41 Depth=1", "=>This Inner Loop Header: Depth=2").  The STEP loop of a blind-rotate kernel is its
depth-1 loop that contains an `s_barrier`; a depth-2 loop inside it is the loop over the gadget rows behind the first one,
whose trip count follows from l and the kernel (`ROW_LOOP_TRIPS`).  All acyclic paths through one iteration of the step loop
are enumerated (inner loops collapsed and multiplied by their trip count); a wave's path depends on wave-uniform
conditions only (its prime q, its role A / B in the 8-wave form, the priority time slice), so every wave runs ONE of these
paths per step.  Paths with the same VALU counts are merged; the skip of a step whose rotation is zero and paths that differ
in scalar instructions only disappear that way.

Classes (what the issue-cost table prices): mul = v_mul_* / v_mad_* integer multiplies; three_operand = other VALU with three
source operands (v_add3_u32, v_bfe_*, v_lshl_add_*, v_and_or_b32, v_perm_b32 ...); two_operand = the rest.
"""
import argparse
import collections
import json
import re
import sys

MUL = ("v_mad_i64_i32", "v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_i32", "v_mul_hi_u32", "v_mul_u32_u24", "v_mul_i32_i24",
       "v_mad_u32_u24", "v_mad_i32_i24", "v_mul_lo_i32")
THREE = ("v_add3_u32", "v_bfe_i32", "v_bfe_u32", "v_lshl_add_u32", "v_lshl_add_u64", "v_add_lshl_u32", "v_lshl_or_b32",
         "v_and_or_b32", "v_or3_b32", "v_xad_u32", "v_perm_b32", "v_alignbit_b32", "v_alignbyte_b32", "v_bfi_b32",
         "v_med3_i32", "v_med3_u32", "v_min3_i32", "v_max3_i32", "v_min3_u32", "v_max3_u32", "v_sad_u32", "v_cndmask_b32_e64",
         "v_xor3_b32")

# kernels of interest: mangled-name needle -> (report name, what it is, roles).  A role is a kind of wave of ONE rotation:
# (name, gadget rows it transforms per step as a function of l, how many waves of that role sit on one SIMD when the
# chip is full).  4-wave form: two workgroups per CU, so a SIMD holds the same (q, u) wave of two rotations; 8-wave form:
# one workgroup per CU, a SIMD holds wave A = (q, u, rows 0..l-2) and wave B = (q, u, last row) of one rotation; split
# form at N = 2048: one workgroup per CU, a SIMD holds the two half-transform waves (q, u, h) of one rotation.
KERNELS = {
    "blind_rotate4_kernelILi10ELb1E": ("blind_rotate4_kernel<10,true>", "4-wave form, digit tables (P128: the headline)",
                                           [("wave", lambda l: l, 2)]),
    "blind_rotate4_kernelILi10ELb0E": ("blind_rotate4_kernel<10,false>", "4-wave form, no digit table (P80)",
                                           [("wave", lambda l: l, 2)]),
    "blind_rotate8_kernelILi10ELb1E": ("blind_rotate8_kernel<10,true>", "8-wave latency form (narrow launches)",
                                       [("A", lambda l: l - 1, 1), ("B", lambda l: 1, 1)]),
    "blind_rotate_split_kernelILi11ELi2E": ("blind_rotate_split_kernel<11,2>", "split form (P2048, BASELINE configs[4])",
                                            [("wave", lambda l: l, 2)]),
}


def classify(op):
    if op in MUL:
        return "mul"
    if op in THREE or (op.endswith("_e64") and op[:-4] in THREE):
        return "three_operand"
    return "two_operand"


def parse_kernel(lines, needle):
    start = next(i for i, l in enumerate(lines) if needle in l and re.match(r"^_Z\w+:", l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    func = re.match(r"^(_Z\w+):", lines[start]).group(1)
    blocks, order = {}, []
    cur = {"name": "entry", "ins": [], "header": None, "depth": 0, "is_header": False}
    for raw in lines[start + 1:end + 1]:
        m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", raw)
        if m:
            blocks[cur["name"]] = cur
            order.append(cur["name"])
            cur = {"name": m.group(1), "ins": [], "header": None, "depth": 0, "is_header": False, "parent": None}
            note = m.group(2) or ""
            h = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", note)
            if h:
                cur["header"], cur["depth"] = ".L" + h.group(1), int(h.group(2))
            h = re.search(r"This (?:Inner )?Loop Header: Depth=(\d+)", note)
            if h:
                cur["header"], cur["depth"], cur["is_header"] = cur["name"], int(h.group(1)), True
            h = re.search(r"Parent Loop (BB\d+_\d+) Depth=(\d+)", note)
            if h:
                cur["parent"] = ".L" + h.group(1)
            continue
        s = raw.strip()
        if s.startswith(";"):
            # continuation lines of a block's loop note: "; Parent Loop BB20_41 Depth=1", "; =>This Inner Loop Header: Depth=2"
            h = re.search(r"Parent Loop (BB\d+_\d+) Depth=(\d+)", s)
            if h and not cur["ins"]:
                cur["parent"] = ".L" + h.group(1)
            h = re.search(r"This (?:Inner )?Loop Header: Depth=(\d+)", s)
            if h and not cur["ins"]:
                cur["header"], cur["depth"], cur["is_header"] = cur["name"], int(h.group(1)), True
            continue
        if s and not s.startswith((".", "//")):
            cur["ins"].append(s)
            if s.split()[0].startswith(("s_cbranch", "s_branch")):
                # a basic block ends at a branch; what follows up to the next label has no label of its own (only branch
                # targets get one): name it after the labelled block, same loop
                blocks[cur["name"]] = cur
                order.append(cur["name"])
                stem, _, k = cur["name"].partition("+")
                cur = dict(cur, name=f"{stem}+{int(k or 0) + 1}", ins=[], is_header=False)
    blocks[cur["name"]] = cur
    order.append(cur["name"])
    return func, blocks, order


def block_counts(ins):
    c = collections.Counter()
    for i in ins:
        op = i.split()[0]
        if op.startswith("v_"):
            c["valu"] += 1
            c[classify(op)] += 1
        elif op.startswith("ds_"):
            c["lds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_")):
            c["vmem"] += 1
        elif op == "s_barrier":
            c["barrier"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
    return c


def successors(blocks, order, name):
    """(taken targets, falls through?) of a block: conditional branches may fall through, s_branch / s_endpgm do not."""
    b = blocks[name]
    targets, falls = [], True
    for i in b["ins"]:
        op = i.split()[0]
        if op.startswith("s_cbranch"):
            targets.append(i.split()[-1])
        elif op in ("s_branch",):
            targets.append(i.split()[-1])
            falls = False
        elif op in ("s_endpgm", "s_setpc_b64"):
            falls = False
    nxt = order.index(name) + 1
    if falls and nxt < len(order):
        targets.append(order[nxt])
    seen, out = set(), []
    for t in targets:
        if t not in seen:
            seen.add(t)
            out.append(t)
    return out


def is_row_body(c):
    """A gadget row of the forward phase: a transform's multiplies and the loads of the row's key image, no barrier."""
    return c["mul"] >= 150 and c["vmem"] >= 8 and c["barrier"] == 0


def analyse(lines, needle, l):
    report, what, roles = KERNELS[needle]
    func, blocks, order = parse_kernel(lines, needle)
    counts = {n: block_counts(blocks[n]["ins"]) for n in order}
    tags = {}                                              # block -> {key: value} from "; isa_mix role key=value" comments
    for n in order:
        for i in blocks[n]["ins"]:
            for k, v in re.findall(r"isa_mix role (\w+)=(\w+)", i):
                tags.setdefault(n, {})[k] = v

    # loop membership: a depth-2 block names its inner header; its depth-1 loop is that header's parent
    def outer_header(n):
        b = blocks[n]
        if b["depth"] == 1:
            return b["header"]
        if b["depth"] == 2:
            return blocks[b["header"]].get("parent")
        return None
    loops1 = collections.defaultdict(list)
    for n in order:
        h = outer_header(n)
        if h:
            loops1[h].append(n)
    step = [h for h, members in loops1.items() if any(counts[m]["barrier"] for m in members)]
    if len(step) != 1:
        raise SystemExit(f"{report}: expected one depth-1 loop with a barrier, found {step}")
    header = step[0]
    members = set(loops1[header])

    # collapse depth-2 loops into one node each (named after the inner header); its trip count is solved per role below
    def node_of(n):
        b = blocks[n]
        return b["header"] if b["depth"] == 2 else n
    node_blocks = collections.defaultdict(list)
    for n in order:
        if n in members:
            node_blocks[node_of(n)].append(n)
    node_cost, node_tags = {}, {}
    for node, bl in node_blocks.items():
        c = collections.Counter()
        for n in bl:
            c.update(counts[n])
            node_tags.setdefault(node, {}).update(tags.get(n, {}))
        node_cost[node] = c
    is_loop = {node: blocks[node]["depth"] == 2 for node in node_blocks}
    edges = collections.defaultdict(list)
    for n in order:
        if n not in members:
            continue
        for t in successors(blocks, order, n):
            if t not in members:
                continue                                   # leaves the step loop (the last iteration)
            a, b = node_of(n), node_of(t)
            if a != b and b not in edges[a]:               # (a == b: the inner loop's own back edge)
                edges[a].append(b)

    paths = []                                             # every path header -> ... -> header: one iteration

    def walk(node, seen, acc):
        acc = acc + [node]
        for t in edges[node]:
            if t == header:
                paths.append(acc)
            elif t not in seen:
                walk(t, seen | {t}, acc)
    walk(header, {header}, [])
    role_keys = {k for t in node_tags.values() for k in t}

    out_roles = []
    for name, rows_of, per_simd in roles:
        rows = rows_of(l)
        found = {}
        for p in paths:
            # exactly one value of every role key marked in the source (the branches of `if (q == 0) ... else ...`)
            seen_tags = collections.defaultdict(set)
            for node in p:
                for k, v in node_tags.get(node, {}).items():
                    seen_tags[k].add(v)
            if any(len(seen_tags[k]) != 1 for k in role_keys):
                continue
            fixed = sum(1 for node in p if not is_loop[node] and is_row_body(node_cost[node]))
            loops = [node for node in p if is_loop[node] and is_row_body(node_cost[node])]
            if len(loops) > 1:
                continue
            trips = rows - fixed if loops else 0
            if (loops and trips < 1) or (not loops and fixed != rows):
                continue
            c = collections.Counter()
            for node in p:
                w = trips if node in loops else 1
                for k, v in node_cost[node].items():
                    c[k] += v * w
            if c["barrier"] == 0:
                continue                                   # the skip of a step whose rotation is zero
            key = (c["valu"], c["mul"], c["three_operand"], c["two_operand"])
            if key not in found:
                found[key] = {"valu": c["valu"], "mul": c["mul"], "three_operand": c["three_operand"], "two_operand": c["two_operand"],
                              "lds": c["lds"], "vmem": c["vmem"], "barriers": c["barrier"], "row_loop_trips": trips,
                              "tags": {k: sorted(v)[0] for k, v in seen_tags.items()}, "blocks": p}
        variants = sorted(found.values(), key=lambda r: -r["valu"])
        if not variants:
            raise SystemExit(f"{report}: no feasible path for role {name} ({rows} gadget rows)")
        out_roles.append({"role": name, "gadget_rows": rows, "waves_per_simd": per_simd, "variants": variants})
    return {"kernel": report, "what": what, "symbol": func, "l": l, "step_loop_header": header, "roles": out_roles}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("asm")
    ap.add_argument("--l", type=int, default=3, help="gadget length (P128 and P2048: 3; P80: 2)")
    ap.add_argument("--json")
    ap.add_argument("--sha", help="kernels_sha16 of the sources the listing was compiled from (peba1_amd/kernel_id.py); default: computed now")
    args = ap.parse_args()
    lines = open(args.asm).read().split("\n")
    out = {}
    for needle, (report, _what, _t) in KERNELS.items():
        l = 2 if "ELb0E" in needle and "rotate4" in needle else args.l
        try:
            out[report] = analyse(lines, needle, l)
        except StopIteration:
            continue
    for name, k in out.items():
        print(f"{name}  (l = {k['l']})")
        for r in k["roles"]:
            for v in r["variants"]:
                print(f"   role {r['role']:5s} x{r['waves_per_simd']} per SIMD, {r['gadget_rows']} rows {v['tags'] or ''}: valu {v['valu']:5d}  mul {v['mul']:4d}  "
                      f"three-operand {v['three_operand']:3d}  two-operand {v['two_operand']:4d}  lds {v['lds']:3d}  vmem {v['vmem']:3d}  "
                      f"barriers {v['barriers']}")
    if args.json:
        sha = args.sha
        if not sha:
            import os
            sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
            from peba1_amd.kernel_id import kernels_sha16
            sha = kernels_sha16()
        doc = {"what": "static VALU instruction mix of ONE blind-rotate step per wave role, from the assembly listing of the kernels as "
                       "built (tools/isa_mix.py; regenerated by __graft_entry__.build() when the kernel sources or flags change)",
               "kernels_sha16": sha, "kernels": out}
        with open(args.json, "w") as f:
            json.dump(doc, f, indent=1)
            f.write("\n")
    return out


if __name__ == "__main__":
    main()
