"""Static check of the shipped gfx950 code for the one hazard the compiler cannot see: the double-precision DPP multiply-adds
of cpi_factor_kernels.hpp (dpp_fmac) are inline assembly, and hipcc's hazard recogniser does not look inside inline
assembly.  CDNA3/4 ISA guide, "manually inserted wait states": a VALU instruction that writes a VGPR followed by a DPP read of
that VGPR needs two wait states.  This script disassembles every code object in the library and, for every *_dpp instruction,
checks that neither of the two preceding instruction slots is a VALU write of the DPP source (src0); s_nop N counts N + 1.

    python tests/tools/dpp_hazards.py [path/to/libcpi_amd.so]      -> prints the counts, exit code 1 on a violation
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib, tmp):
    """The gfx950 ELF of every translation unit's fat binary in the library's .hip_fatbin section."""
    fat = os.path.join(tmp, "fat.bin")
    subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib], check=True)
    blob = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
    out = []
    for i, s in enumerate(starts):
        piece = os.path.join(tmp, "bundle%d.bin" % i)
        with open(piece, "wb") as f:
            f.write(blob[s:starts[i + 1] if i + 1 < len(starts) else len(blob)])
        co = os.path.join(tmp, "co%d.elf" % i)
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + piece,
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True)
        out.append(co)
    return out


def regs(op):
    """v[a:b] / vN -> set of VGPR numbers (empty for anything else)."""
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", op)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", op)
    return {int(m.group(1))} if m else set()


def scan(dis):
    """llvm-objdump -d text -> (number of DPP instructions, [(function, the VALU write, the DPP read)])."""
    total, bad = 0, []
    func, window = None, []        # window: (wait states it occupies, VGPRs it writes as a VALU op, text)
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            func, window = m.group(1), []
            continue
        text = line.split("//")[0].strip()
        if not text or func is None:
            continue
        parts = text.split(None, 1)
        op = parts[0]
        args = [a.strip() for a in parts[1].split(",")] if len(parts) > 1 else []
        if "_dpp" in op:
            total += 1
            src0 = regs(args[1].split()[0]) if len(args) > 1 else set()
            slots = 0
            for states, writes, t in reversed(window):
                if slots >= 2:
                    break
                if writes & src0:
                    bad.append((func, t, text))
                slots += states
        if op == "s_nop":
            window.append((int(args[0], 0) + 1, set(), text))
        else:
            window.append((1, regs(args[0].split()[0]) if (op.startswith("v_") and args) else set(), text))
        window = window[-4:]
    return total, bad


def check(lib):
    total, bad = 0, []
    with tempfile.TemporaryDirectory() as tmp:
        for co in code_objects(lib, tmp):
            dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], check=True,
                                 stdout=subprocess.PIPE, text=True).stdout
            n, b = scan(dis)
            total += n
            bad += b
    return total, bad


if __name__ == "__main__":
    here = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "cpi_amd", "libcpi_amd.so")
    n, bad = check(lib)
    print("%d DPP instructions, %d with a source written by one of the two preceding VALU slots" % (n, len(bad)))
    for f, w, r in bad[:20]:
        print("  %s:\n      %s\n      %s" % (f, w, r))
    sys.exit(1 if bad else 0)
