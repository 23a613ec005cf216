#!/usr/bin/env python3
"""Vector-instruction mix of ONE kernel's emitted gfx950 ISA, priced with the per-opcode issue costs tools/valu_rate.hip measured
on this chip (profiles/r03_valu_rate.txt, profiles/r05_valu_rate.txt: cycles per wave64 instruction per SIMD at 8 waves per SIMD,
independent chains).

  python3 tools/isa_mix.py orb_slam3_v1.0_amd/csrc/kernels_fast.hip fast_blur_kernelILi3E

compiles the file for the device only (`hipcc --cuda-device-only -S`, the library's own flags), takes the function whose mangled
name contains the given substring, counts its `v_*` instructions per opcode and prints one JSON object: the static count, the
count per cost group, and `valu_cycles_per_instruction_weighted` = sum(count x cost) / count.  The mix is STATIC (every basic
block once); the dynamic count per wave comes from the SQ_INSTS_VALU counter (tools/profile_report.py), and the bench line prices
that dynamic count at this static mean -- exact for straight-line code, an estimate where blocks are skipped or looped.
Opcodes the rate tool never measured are priced at the slow group's 4.2 and listed (round 5: none left in fast_blur_kernel; the
64-bit integer forms measured 4.1-5.0 cycles, v_cndmask_b32 4.25)."""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math"]  # csrc/Makefile CXXFLAGS

# measured (profiles/r03_valu_rate.txt); key = mnemonic without the _e32 / _e64 / _sdwa / _dpp suffix
COST = {
    "v_add_u32": 2.58, "v_sub_u32": 2.30, "v_subrev_u32": 2.30, "v_and_b32": 2.56, "v_or_b32": 2.37, "v_xor_b32": 2.32,
    "v_not_b32": 2.22, "v_lshrrev_b32": 2.13, "v_ashrrev_i32": 2.22, "v_mov_b32": 2.21, "v_max_u16": 2.28, "v_min_u16": 2.32,
    "v_add_u16": 2.30, "v_sub_u16": 2.24, "v_min_i16": 2.15,
    # round 5 (profiles/r05_valu_rate.txt): the opcodes that were priced by assumption before
    "v_cndmask_b32": 4.25, "v_max_i16": 2.28, "v_lshlrev_b16": 2.06, "v_lshrrev_b16": 2.10, "v_mul_lo_u16": 2.31,
    "v_readfirstlane_b32": 4.26, "v_lshl_add_u64": 4.56, "v_mad_u64_u32": 4.98, "v_mad_i64_i32": 4.87, "v_mov_b64": 4.19,
    "v_lshlrev_b64": 4.09, "v_add_f32": 2.33, "v_sub_f32": 2.33, "v_mul_f32": 2.30, "v_cvt_u32_f32": 4.09, "v_readlane_b32": 4.28,
    "v_fma_f32": 4.83, "v_min_u32": 4.53, "v_max_u32": 4.53, "v_max3_u32": 4.54, "v_min3_i32": 4.38, "v_med3_u32": 4.23,
    "v_pk_max_u16": 4.20, "v_pk_min_i16": 4.17, "v_pk_add_u16": 4.17, "v_pk_sub_i16": 4.15, "v_pk_mad_u16": 4.31,
    "v_perm_b32": 4.22, "v_alignbyte_b32": 4.16, "v_dot4_u32_u8": 4.33, "v_dot2_u32_u16": 4.20, "v_lshl_or_b32": 4.12,
    "v_and_or_b32": 4.21, "v_bfe_u32": 4.10, "v_bfi_b32": 4.22, "v_mad_u32_u24": 4.15, "v_mad_i32_i24": 4.16,
    "v_sad_u8": 4.20, "v_msad_u8": 4.17, "v_lerp_u8": 4.16, "v_mbcnt_lo_u32_b32": 4.19, "v_mbcnt_hi_u32_b32": 4.19,
    "v_bcnt_u32_b32": 4.16, "v_or3_b32": 4.29, "v_add3_u32": 4.24, "v_mul_lo_u32": 4.15, "v_cvt_f32_ubyte0": 4.09,
    "v_lshlrev_b32": 4.09, "v_max_i32": 4.19, "v_min_i32": 4.19, "v_lshl_add_u32": 4.14, "v_add_lshl_u32": 4.19,
    "v_xad_u32": 4.16, "v_mul_u32_u24": 4.10, "v_mul_i32_i24": 4.10,
}
ASSUMED_FAST = {}
SLOW_DEFAULT, WIDE_DEFAULT = 4.2, 4.6


def base(mn):
    return re.sub(r"_(e32|e64|sdwa|dpp|e64_dpp)$", "", mn)


def cost_of(mn):
    b = base(mn)
    if b in COST:
        return COST[b], "measured"
    if b in ASSUMED_FAST:
        return ASSUMED_FAST[b], "assumed"
    if b.startswith("v_cmp"):
        return 4.5, "measured"  # cmp_only 4.54, cmp_e64 4.42, v_cmp_lt_i16 4.49 / 4.43
    if re.search(r"(_b64|_u64|_i64|i64_i32|u64_u32)$", b):
        return WIDE_DEFAULT, "assumed"  # (the 64-bit forms measured so far run at 4.1-5.0)
    return SLOW_DEFAULT, "assumed"


def kernel_asm(hip_file, symbol_part):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    asm = subprocess.run([hipcc, "--offload-arch=gfx950", *FLAGS, "--cuda-device-only", "-S", hip_file, "-o", "-"],
                         check=True, capture_output=True, text=True).stdout
    out, on, name = [], False, None
    for ln in asm.splitlines():
        m = re.match(r"^(_Z\w+):", ln)
        if m and symbol_part in m.group(1) and not on:
            on, name = True, m.group(1)
            continue
        if on and re.match(r"^\.Lfunc_end\d+:", ln):
            break
        if on:
            out.append(ln)
    if name is None:
        raise SystemExit("no function containing %r in %s" % (symbol_part, hip_file))
    return name, out


def mix(hip_file, symbol_part):
    name, lines = kernel_asm(hip_file, symbol_part)
    counts = {}
    for ln in lines:
        m = re.match(r"^\s+(v_\w+)", ln)
        if m:
            counts[m.group(1)] = counts.get(m.group(1), 0) + 1
    n = sum(counts.values())
    tot = fast = slow = assumed = 0.0
    unmeasured = {}
    for mn, c in counts.items():
        k, how = cost_of(mn)
        tot += k * c
        if k < 3.0:
            fast += c
        else:
            slow += c
        if how == "assumed":
            assumed += c
            unmeasured[base(mn)] = unmeasured.get(base(mn), 0) + c
    return {"symbol": name, "valu_static": n, "valu_static_fast_group": int(fast), "valu_static_slow_group": int(slow),
            "valu_static_priced_by_assumption": int(assumed), "unmeasured_opcodes": dict(sorted(unmeasured.items())),
            "valu_cycles_per_instruction_weighted": tot / n if n else None,
            "cost_table": "profiles/r03_valu_rate.txt + r05_valu_rate.txt (tools/valu_rate.hip); fast group < 3 cycles, slow group ~4.2",
            "salu_static": sum(1 for ln in lines if re.match(r"^\s+s_(?!waitcnt|nop|endpgm|barrier|cbranch|branch|sleep)", ln)),
            "lds_static": sum(1 for ln in lines if re.match(r"^\s+ds_", ln))}


if __name__ == "__main__":
    print(json.dumps(mix(sys.argv[1], sys.argv[2]), indent=1))
