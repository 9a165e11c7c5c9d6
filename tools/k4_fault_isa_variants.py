"""ISA-level bisection of the kernel-4 wrong-bits fault (tools/k4_fault_repro.md).

The failing build (tools/fault_wt = commit 1593180 + k4_fault_repro.patch, CS_EXP=0) is compiled to gfx950
assembly once; every variant is that assembly with ONE edit inside the slot loops of the failing
instantiation cs_propagate_ne_regs<uchar,1,1,2,true>, re-assembled, bundled and linked into
tools/fault_wt/variants/libcsolve_hip_<name>.so (host object compiled around the edited code object with
-fcuda-include-gpubinary).  Built here without a GPU; tools/k4_fault_run_variants.sh runs
tools/fault_wt/tools/validate_k4.py once per library on the GPU box.

usage: python tools/k4_fault_isa_variants.py            (build all)
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WT = os.path.join(ROOT, "tools", "fault_wt")
SRC = os.path.join(WT, "csolve_amd", "csrc")
OUT = os.path.join(WT, "variants")
LLVM = "/opt/rocm/lib/llvm/bin"
KERNEL = "_Z20cs_propagate_ne_regsIhLi1ELi1ELi2ELb1EEviPKT_iiPKiS4_PK6cs_valPKyPK10cs_node_inPS5_PyP11cs_node_outx"
FLAGS = ["-O3", "-g", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-DCS_EXP=0"]


def run(cmd, **kw):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, **kw)
    if r.returncode != 0:
        sys.exit("FAILED: " + " ".join(cmd) + "\n" + r.stdout)
    return r.stdout


def kernel_span(text):
    """(start, end) character offsets of the failing instantiation's body: label .. s_endpgm."""
    m = re.search(r"^" + re.escape(KERNEL) + r":.*$", text, flags=re.M)
    assert m, "kernel label not found"
    e = text.index("s_endpgm", m.end())
    return m.end(), e


# the two copies (D = 2) of the slot loop have this shape; group names are reused by the edits
PAIR = re.compile(
    r"(?P<s1>\tv_lshlrev_b64 v\[\d+:\d+\], v\d+, 1\n)"
    r"(?P<s2>\tv_lshlrev_b64 v\[\d+:\d+\], v\d+, 1\n)"
    r"(?P<c1>\tv_cmp_gt_u32_e32 vcc, 64, v\d+\n)"
    r"(?P<c2>\tv_cmp_gt_u32_e64 s\[(?P<sa>\d+):(?P<sb>\d+)\], 64, v\d+\n)"
    r"(?P<sa3>\ts_add_i32 s\d+, s\d+, 3\n)")
FIRST = re.compile(r"(?P<s0>\tv_lshlrev_b64 v\[\d+:\d+\], v\d+, 1\n)(?P<c0>\tv_cmp_gt_u32_e32 vcc, 64, v\d+\n)(?P<w>\ts_waitcnt lgkmcnt\(1\)\n)")


def edit_pair(body, fn):
    out, n = PAIR.subn(fn, body)
    assert n == 2, f"expected the pattern in both copies of the slot loop, found {n}"
    return out


VARIANTS = {}


def variant(f):
    VARIANTS[f.__name__[2:]] = f
    return f


@variant
def v_A_control(body):
    """no edit: the compiler's code through this flow -- must fail like the plain build"""
    return body


@variant
def v_B_nop_after_compares(body):
    """s_nop 1 after the two compares: compare -> select distance +2 (and shift -> select +2)"""
    return edit_pair(body, lambda m: m["s1"] + m["s2"] + m["c1"] + m["c2"] + "\ts_nop 1\n" + m["sa3"])


@variant
def v_C_nop_around_shifts(body):
    """s_nop 1 between the two 64-bit shifts and after the second"""
    return edit_pair(body, lambda m: m["s1"] + "\ts_nop 1\n" + m["s2"] + "\ts_nop 1\n" + m["c1"] + m["c2"] + m["sa3"])


@variant
def v_D_nop_between_shifts(body):
    """s_nop 0 between the two back-to-back 64-bit shifts only"""
    return edit_pair(body, lambda m: m["s1"] + "\ts_nop 0\n" + m["s2"] + m["c1"] + m["c2"] + m["sa3"])


@variant
def v_E_nop_after_shifts(body):
    """s_nop 1 after the second 64-bit shift only (shift -> compare distance)"""
    return edit_pair(body, lambda m: m["s1"] + m["s2"] + "\ts_nop 1\n" + m["c1"] + m["c2"] + m["sa3"])


@variant
def v_F_valu_filler(body):
    """the SALU filler between compare and select replaced by a VALU one (s_add moved in front of the shifts):
    same instruction distances, but both wait states are vector instructions"""
    return edit_pair(body, lambda m: m["sa3"] + m["s1"] + m["s2"] + m["c1"] + m["c2"] + "\tv_nop\n")


@variant
def v_H_compares_first(body):
    """compares moved in front of the shifts: compare -> select distance +2, shift -> select distance -2"""
    return edit_pair(body, lambda m: m["c1"] + m["c2"] + m["s1"] + m["s2"] + m["sa3"])


@variant
def v_I_other_sgpr_pair(body):
    """the e64 compare and its selects use s[36:37], a pair the scalar unit never writes in this kernel"""
    def fn(m):
        return m["s1"] + m["s2"] + m["c1"] + m["c2"].replace(f"s[{m['sa']}:{m['sb']}]", "s[36:37]") + m["sa3"]
    body = edit_pair(body, fn)
    # the two selects that follow each compare (v_cndmask_b32_e64 ..., s[2:3]) inside the slot loops only
    def fix_loop(mm):
        return mm.group(0).replace("s[2:3]", "s[36:37]")
    body, n = re.subn(r"\tv_cmp_gt_u32_e64 s\[36:37\], 64, v\d+\n(?:.*\n){1,8}?\tv_cndmask_b32_e64 v\d+, 0, v\d+, s\[2:3\]\n\tv_cndmask_b32_e64 v\d+, 0, v\d+, s\[2:3\]\n", fix_loop, body)
    assert n == 2, n
    return body


@variant
def v_G_nop_first_slot(body):
    """s_nop 1 between the FIRST slot's 64-bit shift and its compare"""
    out, n = FIRST.subn(lambda m: m["s0"] + "\ts_nop 1\n" + m["c0"] + m["w"], body)
    assert n == 2, n
    return out


@variant
def v_J_nop_before_shifts(body):
    """s_nop 3 in front of the two 64-bit shifts (lets older vector instructions retire first)"""
    return edit_pair(body, lambda m: "\ts_nop 3\n" + m["s1"] + m["s2"] + m["c1"] + m["c2"] + m["sa3"])


# ---- second series: the LDS reads of the slot loop and the registers that hold their addresses ----------
# copy 1:  ds_read_u8 v27, v26 | ds_read_u8 v43, v43 | ds_read_u8 v46, v44 | s_waitcnt lgkmcnt(2) | v_add | v_sub
#          | v_lshlrev_b64 v[44:45], v27, 1     <- writes v44, the ADDRESS register of the third read
# copy 2:  ds_read_u8 v42, v41 | ds_read_u8 v45, v43 | ds_read_u8 v44, v44 | s_waitcnt lgkmcnt(2) | v_add | v_sub
#          | v_lshlrev_b64 v[42:43], v46, 1     <- writes v43, the ADDRESS register of the second read
READS = re.compile(r"(?P<r2>\tds_read_u8 v\d+, v\d+\n)(?P<r3>\tds_read_u8 v\d+, v\d+\n)(?P<w>\ts_waitcnt lgkmcnt\(2\)\n)")


def edit_reads(body, fn):
    out, n = READS.subn(fn, body)
    assert n == 2, n
    return out


@variant
def v_K_wait_all_reads(body):
    """s_waitcnt lgkmcnt(0) after the third table read: every LDS read has returned before any vector instruction"""
    return edit_reads(body, lambda m: m["r2"] + m["r3"] + "\ts_waitcnt lgkmcnt(0)\n")


@variant
def v_L_rename_shift_dst(body):
    """NO timing change: the first slot's 64-bit shift writes a fresh register pair v[48:49] instead of the pair
    that contains the address register of a table read still in flight"""
    a = ("\tv_lshlrev_b64 v[44:45], v27, 1\n", "\tv_lshlrev_b64 v[48:49], v27, 1\n",
         "\tv_cndmask_b32_e32 v27, 0, v45, vcc\n", "\tv_cndmask_b32_e32 v27, 0, v49, vcc\n",
         "\tv_cndmask_b32_e32 v44, 0, v44, vcc\n\tv_sub_u32_e32 v47, s11, v45\n",
         "\tv_cndmask_b32_e32 v44, 0, v48, vcc\n\tv_sub_u32_e32 v47, s11, v45\n")
    b = ("\tv_lshlrev_b64 v[42:43], v46, 1\n", "\tv_lshlrev_b64 v[48:49], v46, 1\n",
         "\tv_cndmask_b32_e32 v43, 0, v43, vcc\n\tv_cndmask_b32_e32 v42, 0, v42, vcc\n",
         "\tv_cndmask_b32_e32 v43, 0, v49, vcc\n\tv_cndmask_b32_e32 v42, 0, v48, vcc\n")
    for old, new in list(zip(a[0::2], a[1::2])) + list(zip(b[0::2], b[1::2])):
        assert body.count(old) == 1, (old, body.count(old))
        body = body.replace(old, new)
    return body


@variant
def v_M_nop7_after_reads(body):
    """timing only: s_nop 7 between the third table read and the s_waitcnt"""
    return edit_reads(body, lambda m: m["r2"] + m["r3"] + "\ts_nop 7\n" + m["w"])


@variant
def v_N_nop1_after_reads(body):
    """timing only: s_nop 1 between the third table read and the s_waitcnt"""
    return edit_reads(body, lambda m: m["r2"] + m["r3"] + "\ts_nop 1\n" + m["w"])


# ---- third series: what about the renaming of series two made the kernel exact? ---------------------------
def _rename_copy1(body):
    for old, new in (("\tv_lshlrev_b64 v[44:45], v27, 1\n", "\tv_lshlrev_b64 v[48:49], v27, 1\n"),
                     ("\tv_cndmask_b32_e32 v27, 0, v45, vcc\n", "\tv_cndmask_b32_e32 v27, 0, v49, vcc\n"),
                     ("\tv_cndmask_b32_e32 v44, 0, v44, vcc\n\tv_sub_u32_e32 v47, s11, v45\n",
                      "\tv_cndmask_b32_e32 v44, 0, v48, vcc\n\tv_sub_u32_e32 v47, s11, v45\n")):
        assert body.count(old) == 1, old
        body = body.replace(old, new)
    return body


def _rename_copy2(body):
    for old, new in (("\tv_lshlrev_b64 v[42:43], v46, 1\n", "\tv_lshlrev_b64 v[48:49], v46, 1\n"),
                     ("\tv_cndmask_b32_e32 v43, 0, v43, vcc\n\tv_cndmask_b32_e32 v42, 0, v42, vcc\n",
                      "\tv_cndmask_b32_e32 v43, 0, v49, vcc\n\tv_cndmask_b32_e32 v42, 0, v48, vcc\n")):
        assert body.count(old) == 1, old
        body = body.replace(old, new)
    return body


@variant
def v_O_only_more_vgprs(body):
    """the compiler's instructions unchanged; only the kernel descriptor asks for 52 VGPRs instead of 48"""
    return body


@variant
def v_P_rename_copy1(body):
    """series two's renaming in the first copy of the slot loop only (even nodes of a chunk)"""
    return _rename_copy1(body)


@variant
def v_Q_rename_copy2(body):
    """series two's renaming in the second copy of the slot loop only (odd nodes of a chunk)"""
    return _rename_copy2(body)


@variant
def v_R_rename_and_copy_back(body):
    """the shift writes v[48:49], two v_mov copy the result into the registers the compiler chose: the old
    registers are still overwritten at (almost) the same place, but not by the 64-bit shift"""
    body = body.replace("\tv_lshlrev_b64 v[44:45], v27, 1\n", "\tv_lshlrev_b64 v[48:49], v27, 1\n\tv_mov_b32_e32 v44, v48\n\tv_mov_b32_e32 v45, v49\n")
    body = body.replace("\tv_lshlrev_b64 v[42:43], v46, 1\n", "\tv_lshlrev_b64 v[48:49], v46, 1\n\tv_mov_b32_e32 v42, v48\n\tv_mov_b32_e32 v43, v49\n")
    assert body.count("v[48:49]") == 2
    return body


@variant
def v_S_rename_addresses(body):
    """the other way round: the shifts keep their registers, the table reads whose address register they
    overwrite take their address from fresh registers v48 / v49"""
    for old, new in (("\tv_lshl_add_u32 v44, s3, 6, v8\n\tds_read_u8 v43, v43\n\tds_read_u8 v46, v44\n",
                      "\tv_lshl_add_u32 v48, s3, 6, v8\n\tds_read_u8 v43, v43\n\tds_read_u8 v46, v48\n"),
                     ("\tv_lshl_add_u32 v43, s2, 6, v8\n\tv_lshl_add_u32 v44, s3, 6, v8\n\tds_read_u8 v45, v43\n",
                      "\tv_lshl_add_u32 v49, s2, 6, v8\n\tv_lshl_add_u32 v44, s3, 6, v8\n\tds_read_u8 v45, v49\n")):
        assert body.count(old) == 1, old
        body = body.replace(old, new)
    return body


# ---- fourth series: which property of the descriptor matters? ------------------------------------------------
@variant
def v_T_alloc56_accum48(body):
    """instructions unchanged; descriptor: next_free_vgpr 56 (allocation 56) but accum_offset left at 48"""
    return body


@variant
def v_V_top_of_56(body):
    """descriptor asks for 56 registers AND the kernel's v44..v47 are renamed v52..v55: the same values live in
    the top four registers of the (larger) allocation"""
    def ren(m):
        return "v" + str(int(m.group(1)) + 8)
    body = re.sub(r"\bv(4[4-7])\b", ren, body)
    body = re.sub(r"v\[(4[4-7]):(4[4-7])\]", lambda m: f"v[{int(m.group(1)) + 8}:{int(m.group(2)) + 8}]", body)
    assert "v44" not in body and "v[44:45]" not in body and "v[52:53]" in body
    return body


@variant
def v_W_swap_top_with_constants(body):
    """allocation stays 48; v44..v47 (temporaries of the slot loop: LDS addresses and data, shift results) swap
    names with v36..v39 (per-lane constants: lane, root_lo, column, degree).  The top four registers then only
    HOLD values, the traffic goes through v36..v39."""
    m = {36: 44, 37: 45, 38: 46, 39: 47, 44: 36, 45: 37, 46: 38, 47: 39}
    body = re.sub(r"\bv(3[6-9]|4[4-7])\b", lambda x: "v" + str(m[int(x.group(1))]), body)
    body = re.sub(r"v\[(3[6-9]|4[4-7]):(3[6-9]|4[4-7])\]", lambda x: f"v[{m[int(x.group(1))]}:{m[int(x.group(2))]}]", body)
    return body


def _swap(body, m):
    pat = "|".join(str(k) for k in m)
    body = re.sub(r"\bv(" + pat + r")\b", lambda x: "v" + str(m[int(x.group(1))]), body)
    return re.sub(r"v\[(" + pat + r"):(" + pat + r")\]", lambda x: f"v[{m[int(x.group(1))]}:{m[int(x.group(2))]}]", body)


@variant
def v_WA_swap_44_45(body):
    """as W, but only v44, v45 swap with v36, v37: v46, v47 stay temporaries in the top four"""
    return _swap(body, {36: 44, 37: 45, 44: 36, 45: 37})


@variant
def v_WB_swap_46_47(body):
    """as W, but only v46, v47 swap with v38, v39: v44, v45 stay temporaries in the top four"""
    return _swap(body, {38: 46, 39: 47, 46: 38, 47: 39})


@variant
def v_VT_tripwire(body):
    """variant V (fails: temporaries in v52..v55 = top four of 56) plus a tripwire in the first copy of the slot
    loop: the third table read is issued twice, once into v54 as compiled and once into the free register v48;
    where the two differ at the point of use, 2^20 is added to the node's revision count."""
    body = VARIANTS["V_top_of_56"](body)
    old = ("\tds_read_u8 v54, v52\n\ts_waitcnt lgkmcnt(2)\n\tv_add_u32_e32 v27, s13, v27\n\tv_sub_u32_e32 v27, s11, v27\n"
           "\tv_lshlrev_b64 v[52:53], v27, 1\n\tv_cmp_gt_u32_e32 vcc, 64, v27\n\ts_waitcnt lgkmcnt(1)\n"
           "\tv_add_u32_e32 v43, s13, v43\n\tv_sub_u32_e32 v43, s11, v43\n\tv_cndmask_b32_e32 v27, 0, v53, vcc\n"
           "\ts_waitcnt lgkmcnt(0)\n")
    new = ("\tds_read_u8 v54, v52\n\tds_read_u8 v48, v52\n\ts_waitcnt lgkmcnt(3)\n\tv_add_u32_e32 v27, s13, v27\n\tv_sub_u32_e32 v27, s11, v27\n"
           "\tv_lshlrev_b64 v[52:53], v27, 1\n\tv_cmp_gt_u32_e32 vcc, 64, v27\n\ts_waitcnt lgkmcnt(2)\n"
           "\tv_add_u32_e32 v43, s13, v43\n\tv_sub_u32_e32 v43, s11, v43\n\tv_cndmask_b32_e32 v27, 0, v53, vcc\n"
           "\ts_waitcnt lgkmcnt(0)\n"
           "\tv_cmp_ne_u32_e64 s[36:37], v48, v54\n\ts_cmp_lg_u64 s[36:37], 0\n\ts_cselect_b32 s36, 0x100000, 0\n\ts_add_i32 s31, s31, s36\n")
    assert body.count(old) == 1
    return body.replace(old, new)


def main():
    os.makedirs(OUT, exist_ok=True)
    dev_s = os.path.join(OUT, "dev_exp0.s")
    # no -g for the device side: same instructions, no .loc / .Ltmp lines between them
    run(["/opt/rocm/bin/hipcc", *[f for f in FLAGS if f != "-g"], "-S", "--cuda-device-only", "cs_capi.hip", "-o", dev_s],
        cwd=SRC)
    text = open(dev_s).read()
    a, b = kernel_span(text)
    names = sys.argv[1:] or list(VARIANTS)
    for name in names:
        body = VARIANTS[name](text[a:b])
        t = text[:a] + body + text[b:]
        if name.startswith("VT"):
            t = re.sub(r"(\.amdhsa_kernel " + re.escape(KERNEL) + r"\n(?:.*\n)*?\t\t\.amdhsa_next_free_sgpr )36", r"\g<1>38", t, count=1)
        if name[0] in "TV":
            nv, ao = ("56", "48") if name[0] == "T" else ("56", "56")
            t = re.sub(r"(\.amdhsa_kernel " + re.escape(KERNEL) + r"\n(?:.*\n)*?\t\t\.amdhsa_next_free_vgpr )48", r"\g<1>" + nv, t, count=1)
            t = re.sub(r"(\.amdhsa_kernel " + re.escape(KERNEL) + r"\n(?:.*\n)*?\t\t\.amdhsa_accum_offset )48", r"\g<1>" + ao, t, count=1)
            assert f".amdhsa_next_free_vgpr {nv}" in t
        if name[0] in "LOPQRS":
            t = re.sub(r"(\.amdhsa_kernel " + re.escape(KERNEL) + r"\n(?:.*\n)*?\t\t\.amdhsa_next_free_vgpr )48", r"\g<1>52", t, count=1)
            t = re.sub(r"(\.amdhsa_kernel " + re.escape(KERNEL) + r"\n(?:.*\n)*?\t\t\.amdhsa_accum_offset )48", r"\g<1>52", t, count=1)
            assert ".amdhsa_next_free_vgpr 52" in t and ".amdhsa_accum_offset 52" in t
        if name.startswith("I_"):
            t = re.sub(r"(\.amdhsa_kernel " + re.escape(KERNEL) + r"\n(?:.*\n)*?\t\t\.amdhsa_next_free_sgpr )36", r"\g<1>38", t, count=1)
            assert ".amdhsa_next_free_sgpr 38" in t
        s = os.path.join(OUT, f"dev_{name}.s")
        open(s, "w").write(t)
        o, co, fb = s[:-2] + ".o", s[:-2] + ".co", s[:-2] + ".hipfb"
        run([f"{LLVM}/clang", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", s, "-o", o])
        run([f"{LLVM}/ld.lld", "-shared", o, "-o", co])
        run([f"{LLVM}/clang-offload-bundler", "-type=o", "-bundle-align=4096",
             "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950",
             "-input=/dev/null", f"-input={co}", f"-output={fb}"])
        host_o = os.path.join(OUT, f"cs_capi_{name}.o")
        run(["/opt/rocm/bin/hipcc", *FLAGS, "--cuda-host-only", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", fb,
             "-c", "cs_capi.hip", "-o", host_o], cwd=SRC)
        lib = os.path.join(OUT, f"libcsolve_hip_{name}.so")
        objs = [os.path.join(SRC, "build", x) for x in
                ("cs_search.o", "cs_frontend.o", "cs_model.o", "cs_normalize.o", "cs_device.o")]
        run(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-o", lib, host_o, *objs])
        for f in (o, co, fb, host_o):
            os.remove(f)
        print("built", lib)


if __name__ == "__main__":
    main()
