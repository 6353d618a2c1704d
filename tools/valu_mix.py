"""Static instruction mix of the fused rollout kernel's step loops, priced with the per-opcode issue costs measured by
tools/microbench/valu_rate.hip (profiles/*_valu_rate.log).

    hipcc ... --cuda-device-only -S -o pass.s ml2048_hip.hip      (one of build.py's plain passes)
    python tools/valu_mix.py pass.s '<kernel symbol substring>' profiles/r03_valu_rate.log [out.json]

For every depth-2 loop of the kernel (the unrolled step loops) it prints the number of vector instructions per step
(per v_dot4 cluster: eight or nine v_dot4_u32_u8 = one spawn = one step), the share of them that issue at half rate on a
busy SIMD, and the mean cost of one instruction of the mix on a SIMD kept busy by several waves.  The counts are STATIC
(blocks that a wave skips - the re-deal, the Philox refresh - count in full); the dynamic instruction count per step is
what the SQ counters give (profiles/valu_insts.json)."""
import collections
import json
import re
import sys

asm, sym, rate_log = sys.argv[1:4]
out_path = sys.argv[4] if len(sys.argv) > 4 else None

# ---- measured costs: ns per wave-instruction per SIMD with 8 waves per SIMD, ticks per instruction for a lone wave
cost8, lone = {}, {}
for line in open(rate_log):
    m = re.match(r"(\w+)\s+1w:\s+([\d.]+) ns \(\s*([\d.]+) ticks.*8w:\s+([\d.]+) ns", line)
    if m:
        cost8[m.group(1)], lone[m.group(1)] = float(m.group(4)), float(m.group(3))
full_ns = cost8["add_u32"]
alias = {"v_lshl_add_u32": "add3_u32", "v_add_lshl_u32": "add3_u32", "v_mad_u64_u32": "mul_lo_u32", "v_mul_hi_u32": "mul_lo_u32",
         "v_alignbit_b32": "perm_b32", "v_min_u32": "add_u32", "v_max_u32": "add_u32", "v_subrev_u32": "sub_u32",
         "v_cvt_f32_u32": "fma_f32", "v_cvt_f32_i32": "fma_f32", "v_add_f32": "fma_f32", "v_cvt_u32_f32": "fma_f32",
         "v_accvgpr_write_b32": "mov_b32", "v_accvgpr_read_b32": "mov_b32", "v_add_co_u32": "addc_co_u32", "v_ashrrev_i32": "lshrrev_b32"}


def price(op):
    """(ns per instruction on a busy SIMD, known?)"""
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if base.startswith("v_cmp"):
        return cost8["cmp_eq_u32"], True
    if base == "v_cndmask_b32":
        # (in either encoding: the VOP2 form reading a VCC that no VALU instruction wrote recently measures 9.5 ns in
        # the microbenchmark, but the compiler only emits it right behind the v_cmp that produces its mask, where the
        # compare-and-select pair costs 2.64 ns = a half-rate compare + a select at 0.9 ns)
        return cost8["cndmask_b32_e64"], True
    if base in alias:
        return cost8[alias[base]], True
    key = base[2:]
    if op.endswith("_sdwa"):
        return cost8["lshlrev_b32_sdwa"], True
    if key in cost8:
        return cost8[key], True
    return cost8["perm_b32"], False  # unknown: priced at half rate


lines = open(asm).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % re.escape(sym), l))
end = next(i for i in range(start, len(lines)) if ".end_amdhsa_kernel" in lines[i] or lines[i].strip() == "s_endpgm")
loops = collections.defaultdict(collections.Counter)
cur = None
for l in lines[start:end]:
    m = re.search(r"Loop Header: Depth=2|in Loop: Header=(\S+) Depth=2|Parent Loop \S+ Depth=1", l)
    if re.match(r"^\.LBB\w+:", l) or l.lstrip().startswith("; %bb."):
        h = re.search(r"in Loop: Header=(\S+) Depth=2", l)
        if "This Inner Loop Header: Depth=2" in l or ("This Loop Header: Depth=2" in l):
            cur = re.match(r"^\.?L?(BB\w+):", l.replace(".L", "", 1)).group(1) if re.match(r"^\.LBB\w+:", l) else cur
        elif h:
            cur = h.group(1)
        else:
            cur = None
        continue
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        continue
    if cur:
        loops[cur][t.split()[0]] += 1

report = {}
for head, ops in loops.items():
    dots = ops.get("v_dot4_u32_u8", 0)
    if dots < 16:
        continue
    steps = round(dots / (9 if ops.get("ds_read_u16", 0) == 0 and dots % 9 == 0 else 8))
    steps = max(1, round(dots / 8))
    valu = {o: n for o, n in ops.items() if o.startswith("v_")}
    nv = sum(valu.values())
    ns = sum(n * price(o)[0] for o, n in valu.items())
    half = sum(n for o, n in valu.items() if price(o)[0] > 1.3 * full_ns)
    unknown = sorted({o for o in valu if not price(o)[1]})
    report[head] = {
        "steps_in_loop_body": steps, "valu_static_per_step": round(nv / steps, 1), "half_rate_share": round(half / nv, 3),
        "mean_ns_per_inst_busy_simd": round(ns / nv, 4), "mean_cost_vs_full_rate": round(ns / nv / full_ns, 3),
        "lds_reads_per_step": round(ops.get("ds_read_u16", 0) / steps, 2),
        "philox_mads_per_step": round(ops.get("v_mad_u64_u32", 0) / steps, 2),
        "salu_per_step": round(sum(n for o, n in ops.items() if o.startswith("s_") and not o.startswith("s_waitcnt") and not o.startswith("s_nop")) / steps, 1),
        "s_nop_per_step": round(ops.get("s_nop", 0) / steps, 1), "unpriced_opcodes": unknown,
        "top": dict(collections.Counter(valu).most_common(12)),
    }
    print(f"{head}: {steps} steps/body  VALU/step {nv / steps:6.1f}  half-rate share {half / nv:.3f}  "
          f"mean cost {ns / nv / full_ns:.3f} x full rate  LDS reads/step {ops.get('ds_read_u16', 0) / steps:.1f}  "
          f"Philox mads/step {ops.get('v_mad_u64_u32', 0) / steps:.1f}  unpriced {unknown}")
if out_path:
    json.dump({"asm": asm, "kernel": sym, "full_rate_ns": full_ns, "loops": report}, open(out_path, "w"), indent=1)
