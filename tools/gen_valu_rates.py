"""Writes tools/valu_rates.inc for tools/valu_rates.hip: one block of 16 independent instructions with fixed
VGPR numbers per opcode (sources v32..v47, destinations v48..v63), so that the sustained issue rate of each
opcode on gfx950 can be read off a loop that does nothing else."""
import os

OPS = [
    # name, template ({d} destination, {a} {b} {c} sources), lane-operations per instruction
    ("v_add_u32", "v_add_u32 v{d}, v{a}, v{b}"),
    ("v_sub_u32", "v_sub_u32 v{d}, v{a}, v{b}"),
    ("v_and_b32", "v_and_b32 v{d}, v{a}, v{b}"),
    ("v_or_b32", "v_or_b32 v{d}, v{a}, v{b}"),
    ("v_xor_b32", "v_xor_b32 v{d}, v{a}, v{b}"),
    ("v_min_u32", "v_min_u32 v{d}, v{a}, v{b}"),
    ("v_max_u32", "v_max_u32 v{d}, v{a}, v{b}"),
    ("v_min_i32", "v_min_i32 v{d}, v{a}, v{b}"),
    ("v_lshlrev_b32", "v_lshlrev_b32 v{d}, v{a}, v{b}"),
    ("v_lshrrev_b32", "v_lshrrev_b32 v{d}, v{a}, v{b}"),
    ("v_lshlrev_b32 imm", "v_lshlrev_b32 v{d}, 8, v{b}"),
    ("v_mov_b32", "v_mov_b32 v{d}, v{a}"),
    ("v_cndmask_b32", "v_cndmask_b32 v{d}, v{a}, v{b}, vcc"),
    ("v_mul_u32_u24", "v_mul_u32_u24 v{d}, v{a}, v{b}"),
    ("v_mul_lo_u32", "v_mul_lo_u32 v{d}, v{a}, v{b}"),
    ("v_min3_u32", "v_min3_u32 v{d}, v{a}, v{b}, v{c}"),
    ("v_max3_u32", "v_max3_u32 v{d}, v{a}, v{b}, v{c}"),
    ("v_med3_u32", "v_med3_u32 v{d}, v{a}, v{b}, v{c}"),
    ("v_min3_i32", "v_min3_i32 v{d}, v{a}, v{b}, v{c}"),
    ("v_add3_u32", "v_add3_u32 v{d}, v{a}, v{b}, v{c}"),
    ("v_lshl_add_u32", "v_lshl_add_u32 v{d}, v{a}, 2, v{c}"),
    ("v_lshl_or_b32", "v_lshl_or_b32 v{d}, v{a}, 8, v{c}"),
    ("v_and_or_b32", "v_and_or_b32 v{d}, v{a}, v{b}, v{c}"),
    ("v_or3_b32", "v_or3_b32 v{d}, v{a}, v{b}, v{c}"),
    ("v_xad_u32", "v_xad_u32 v{d}, v{a}, v{b}, v{c}"),
    ("v_mad_u32_u24", "v_mad_u32_u24 v{d}, v{a}, v{b}, v{c}"),
    ("v_mad_i32_i24", "v_mad_i32_i24 v{d}, v{a}, v{b}, v{c}"),
    ("v_bfe_u32", "v_bfe_u32 v{d}, v{a}, 8, 8"),
    ("v_bfi_b32", "v_bfi_b32 v{d}, v{a}, v{b}, v{c}"),
    ("v_perm_b32", "v_perm_b32 v{d}, v{a}, v{b}, v{c}"),
    ("v_alignbyte_b32", "v_alignbyte_b32 v{d}, v{a}, v{b}, 1"),
    ("v_alignbit_b32", "v_alignbit_b32 v{d}, v{a}, v{b}, 8"),
    ("v_bcnt_u32_b32", "v_bcnt_u32_b32 v{d}, v{a}, v{b}"),
    ("v_sad_u8", "v_sad_u8 v{d}, v{a}, v{b}, v{c}"),
    ("v_sad_u32", "v_sad_u32 v{d}, v{a}, v{b}, v{c}"),
    ("v_dot4_u32_u8", "v_dot4_u32_u8 v{d}, v{a}, v{b}, v{c}"),
    ("v_dot2_u32_u16", "v_dot2_u32_u16 v{d}, v{a}, v{b}, v{c}"),
    ("v_pk_min_u16", "v_pk_min_u16 v{d}, v{a}, v{b}"),
    ("v_pk_max_u16", "v_pk_max_u16 v{d}, v{a}, v{b}"),
    ("v_pk_min_i16", "v_pk_min_i16 v{d}, v{a}, v{b}"),
    ("v_pk_add_u16", "v_pk_add_u16 v{d}, v{a}, v{b}"),
    ("v_pk_sub_i16", "v_pk_sub_i16 v{d}, v{a}, v{b}"),
    ("v_pk_lshlrev_b16", "v_pk_lshlrev_b16 v{d}, v{a}, v{b}"),
    ("v_pk_mad_u16", "v_pk_mad_u16 v{d}, v{a}, v{b}, v{c}"),
    ("v_cmp_lt_u32 (vcc)", "v_cmp_lt_u32 vcc, v{a}, v{b}"),
    ("v_cmp_lt_u32 (sgpr pair)", "v_cmp_lt_u32 s[20:21], v{a}, v{b}"),
    ("v_add_f32", "v_add_f32 v{d}, v{a}, v{b}"),
    ("v_mul_f32", "v_mul_f32 v{d}, v{a}, v{b}"),
    ("v_fma_f32", "v_fma_f32 v{d}, v{a}, v{b}, v{c}"),
    ("v_mac-like v_fmac_f32", "v_fmac_f32 v{d}, v{a}, v{b}"),
    ("v_min_f32", "v_min_f32 v{d}, v{a}, v{b}"),
    ("v_cvt_f32_u32", "v_cvt_f32_u32 v{d}, v{a}"),
    ("v_cvt_f32_ubyte0", "v_cvt_f32_ubyte0 v{d}, v{a}"),
    ("v_cvt_u32_f32", "v_cvt_u32_f32 v{d}, v{a}"),
    ("v_cvt_pk_u8_f32", "v_cvt_pk_u8_f32 v{d}, v{a}, 0, v{c}"),
    ("v_rcp_f32", "v_rcp_f32 v{d}, v{a}"),
    ("v_add_u32 dpp row_shr:1", "v_add_u32_dpp v{d}, v{a}, v{b} row_shr:1 row_mask:0xf bank_mask:0xf"),
    ("v_mov_b32 dpp quad_perm", "v_mov_b32_dpp v{d}, v{a} quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"),
    ("v_add_u32 sdwa byte", "v_add_u32_sdwa v{d}, v{a}, v{b} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD"),
    ("v_pk_mul_f32", "v_pk_mul_f32 v[{d2}:{d3}], v[{a2}:{a3}], v[{b2}:{b3}]"),
    ("v_pk_add_f32", "v_pk_add_f32 v[{d2}:{d3}], v[{a2}:{a3}], v[{b2}:{b3}]"),
    ("v_add_f64", "v_add_f64 v[{d2}:{d3}], v[{a2}:{a3}], v[{b2}:{b3}]"),
    ("v_mul_f64", "v_mul_f64 v[{d2}:{d3}], v[{a2}:{a3}], v[{b2}:{b3}]"),
    ("v_lshl_add_u64", "v_lshl_add_u64 v[{d2}:{d3}], v[{a2}:{a3}], 2, v[{b2}:{b3}]"),
    ("v_mbcnt_lo_u32_b32", "v_mbcnt_lo_u32_b32 v{d}, v{a}, v{b}"),
    ("v_readlane_b32", "v_readlane_b32 s20, v{a}, 5"),
    ("v_readfirstlane_b32", "v_readfirstlane_b32 s20, v{a}"),
    ("ds_read_b32", "ds_read_b32 v{d}, v{z}"),
    ("ds_read_b128", "ds_read_b128 v[{d4}:{d7}], v{z}"),
    ("ds_read_u8", "ds_read_u8 v{d}, v{z}"),
]


def block(tpl):
    lines = []
    for i in range(16):
        a, b, c, d = 32 + i % 8, 40 + (i + 1) % 8, 32 + (i + 3) % 8, 48 + (i * 2 + 1) % 16
        d2 = 48 + 2 * (i % 8)
        a2, b2 = 32 + 2 * (i % 4), 40 + 2 * ((i + 1) % 4)
        d4 = 48 + 4 * (i % 4)
        lines.append(tpl.format(a=a, b=b, c=c, d=d, d2=d2, d3=d2 + 1, a2=a2, a3=a2 + 1, b2=b2, b3=b2 + 1, d4=d4, d7=d4 + 3, z=31))
    return lines


with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "valu_rates.inc"), "w") as f:
    f.write("// Generated by tools/gen_valu_rates.py.\n")
    f.write("#define RATE_COUNT %d\n" % len(OPS))
    f.write("static const char* const kRateNames[RATE_COUNT] = {%s};\n" % ", ".join('"%s"' % n for n, _ in OPS))
    for k, (name, tpl) in enumerate(OPS):
        f.write("#define RATE_BLK_%d \\\n    %s\n" % (k, " \\\n    ".join('"%s\\n"' % l for l in block(tpl))))
    f.write("#define RATE_DISPATCH(M) \\\n")
    f.write(" \\\n".join("    if (MODE == %d) asm volatile(RATE_BLK_%d ::: RATE_CLOBBERS);" % (k, k) for k in range(len(OPS))))
    f.write("\n#define RATE_RUN_ALL(F) \\\n")
    f.write(" \\\n".join("    F(%d)" % k for k in range(len(OPS))))
    f.write("\n")
