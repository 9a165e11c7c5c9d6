"""ISA lint for the gfx950 hazard behind the kernel-4 fault (tools/k4_fault_repro.md).

On MI355X a 64-bit vector shift (v_lshlrev_b64, v_lshrrev_b64, v_ashrrev_i64) whose 32-bit shift amount sits in
the LAST vector register the wave owns intermittently computes with v0 as the amount (the hardware's
substitute for an out-of-range source register) as soon as several waves share a SIMD
(tools/shift64_last_vgpr.hip reproduces it in isolation: 28 % of such shifts wrong at 8 waves per SIMD, none
with the amount one register lower or with one wave per SIMD).  hipcc 7.2 does not avoid the allocation.

What is checked is the code that SHIPS: the gfx950 code objects are taken out of the built
csolve_amd/libcsolve_hip.so (llvm-objdump --offloading), disassembled, and every kernel's allocation is read
from the code object's own metadata (.vgpr_count of the notes; allocation = that rounded up to the granule of
8).  Register allocation depends on the compiler flags (the Makefile's -mllvm options change it in every
kernel with atomics), so assembly produced with other flags proves nothing about the library.  Without a
built library (or with --from-source) the two translation units are compiled to assembly with the Makefile's
own HIPFLAGS.  Given .s files are checked as they are.  Exit status 1 if a hazardous shift is found.

usage: python tools/check_isa_shift64.py [--from-source] [file.s ...]
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "csolve_amd", "csrc")
LIB = os.path.join(ROOT, "csolve_amd", "libcsolve_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"
UNITS = ("cs_capi.hip", "cs_search.hip")
SHIFT = re.compile(r"^\s*(v_lshlrev_b64|v_lshrrev_b64|v_ashrrev_i64)\s+v\[\d+:\d+\],\s*v(\d+)\s*,")
GRANULE = 8


def makefile_hipflags():
    """the HIPFLAGS line of csolve_amd/csrc/Makefile, $(ARCH) expanded, without -c / -fPIC / -g"""
    text = open(os.path.join(SRC, "Makefile")).read()
    arch = re.search(r"^ARCH\s*\?=\s*(\S+)", text, flags=re.M).group(1)
    flags = re.search(r"^HIPFLAGS\s*\?=\s*(.*)$", text, flags=re.M).group(1).replace("$(ARCH)", arch).split()
    return [f for f in flags if f not in ("-fPIC", "-g", "-c")]


def compile_to_asm(unit):
    out = subprocess.run(["/opt/rocm/bin/hipcc"] + makefile_hipflags() + ["-S", "--cuda-device-only", unit, "-o", "-"],
                         cwd=SRC, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, check=True)
    return out.stdout


def kernels(text):
    """yield (name, body lines, next_free_vgpr) per kernel of one assembly file"""
    alloc = dict(re.findall(r"\.amdhsa_kernel (\S+)\n(?:.*\n)*?\s*\.amdhsa_next_free_vgpr (\d+)", text))
    for m in re.finditer(r"^(\S+):\s*; @\1\n(.*?)\n\s*s_endpgm", text, flags=re.M | re.S):
        name = m.group(1)
        if name in alloc:
            yield name, m.group(2).splitlines(), int(alloc[name])


def check(text, label):
    bad, seen, shifts = [], 0, 0
    for name, lines, nfv in kernels(text):
        seen += 1
        owned = (nfv + GRANULE - 1) // GRANULE * GRANULE
        for ln in lines:
            m = SHIFT.match(ln)
            if m:
                shifts += 1
                if int(m.group(2)) == owned - 1:
                    bad.append((label, name, nfv, ln.strip()))
    return seen, shifts, bad


def shipped_code_objects(lib):
    """the gfx950 code objects bundled in the shared library -> list of (label, path); files live in a temp dir"""
    tmp = tempfile.mkdtemp(prefix="cs_isa_")
    work = os.path.join(tmp, "lib.so")
    shutil.copy(lib, work)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", work], cwd=tmp, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL, check=True)
    objs = sorted(f for f in os.listdir(tmp) if "amdgcn" in f and os.path.getsize(os.path.join(tmp, f)) > 0)
    return tmp, [(f"libcsolve_hip.so[{i}]", os.path.join(tmp, f)) for i, f in enumerate(objs)]


def check_object(path, label):
    """one code object: allocations from its notes, instructions from its disassembly"""
    notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", path], stdout=subprocess.PIPE,
                           stderr=subprocess.DEVNULL, text=True, check=True).stdout
    alloc, name = {}, None
    for ln in notes.splitlines():
        m = re.match(r"\s*\.name:\s+(\S+)", ln)
        if m:
            name = m.group(1)
        m = re.match(r"\s*\.vgpr_count:\s+(\d+)", ln)
        if m and name is not None:
            alloc[name] = int(m.group(1))
    dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", path], stdout=subprocess.PIPE,
                         stderr=subprocess.DEVNULL, text=True, check=True).stdout
    bad, seen, shifts, cur = [], 0, 0, None
    for ln in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", ln)
        if m:
            cur = m.group(1) if m.group(1) in alloc else None
            seen += cur is not None
            continue
        if cur is None:
            continue
        m = SHIFT.match(ln)
        if m:
            shifts += 1
            owned = (alloc[cur] + GRANULE - 1) // GRANULE * GRANULE
            if int(m.group(2)) == owned - 1:
                bad.append((label, cur, alloc[cur], ln.strip()))
    return seen, shifts, bad


def main():
    args = [a for a in sys.argv[1:] if a != "--from-source"]
    from_source = "--from-source" in sys.argv[1:]
    total_k = total_s = 0
    bad = []
    if args:
        what = "given assembly"
        for f in args:
            k, s, b = check(open(f).read(), f)
            total_k, total_s, bad = total_k + k, total_s + s, bad + b
    elif os.path.exists(LIB) and not from_source:
        what = "shipped code objects of csolve_amd/libcsolve_hip.so"
        tmp, objs = shipped_code_objects(LIB)
        try:
            if not objs:
                print("no gfx950 code object found in " + LIB)
                return 1
            for label, path in objs:
                k, s, b = check_object(path, label)
                total_k, total_s, bad = total_k + k, total_s + s, bad + b
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    else:
        what = "sources compiled with the Makefile's HIPFLAGS (" + " ".join(makefile_hipflags()) + ")"
        for u in UNITS:
            k, s, b = check(compile_to_asm(u), u)
            total_k, total_s, bad = total_k + k, total_s + s, bad + b
    print(f"{what}: {total_k} kernels, {total_s} 64-bit shifts with the amount in a vector register, "
          f"{len(bad)} with the amount in the last register of the allocation")
    for label, name, nfv, ln in bad:
        print(f"  {label}: {name} (vgprs {nfv}): {ln}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
