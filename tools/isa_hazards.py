"""Check the device code of libanihip.so for a hazard the compiler cannot see: gfx950 needs two wait states between a VALU
write of a VGPR and an MFMA reading it, and hipcc inserts them for the instructions it schedules itself -- not behind inline
assembly (the v_fma_mix{lo,hi}_f16 pairs of the fp32 -> {hi, lo} fp16 split in csrc/mlp.hip).  The code objects are taken
out of the library's .hip_fatbin section and disassembled with llvm-objdump.

    python tools/isa_hazards.py [path/to/libanihip.so]      -> exit status 1 if a violation is found
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
INLINE_ASM_VALU = ("v_fma_mixlo_f16", "v_fma_mixhi_f16")   # VALU instructions that reach the library through asm()


def code_objects(lib_path):
    """The gfx950 code objects inside the (concatenated) offload bundles of the library."""
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", lib_path], check=True)
        blob = open(fat, "rb").read()
    out, pos = [], blob.find(MAGIC)
    while pos >= 0:
        n, = struct.unpack_from("<Q", blob, pos + len(MAGIC))
        p = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if "gfx950" in triple and size:
                out.append(blob[pos + off:pos + off + size])
        pos = blob.find(MAGIC, pos + 1)
    return out


def instructions(code):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(code)
        f.flush()
        text = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", f.name], check=True,
                              capture_output=True, text=True).stdout
    kernel = None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            kernel = m.group(1)
            continue
        line = line.split("//")[0].strip()
        if line and kernel and re.match(r"^[a-z]", line):
            yield kernel, line


def check(lib_path):
    """(pairs checked, list of violations).  (a) VALU write -> MFMA read: two wait states."""
    checked, bad = 0, []
    for code in code_objects(lib_path):
        ins = list(instructions(code))
        for k, (kern, line) in enumerate(ins):
            if not line.startswith(INLINE_ASM_VALU):
                continue
            # (b) a transcendental result needs a wait state before a VALU instruction reads it (trans forwarding hazard)
            if k > 0 and re.match(r"v_(exp|log|rcp|rsq|sqrt|sin|cos)_", ins[k - 1][1]):
                checked += 1
                tdst = re.match(r"v(\d+)", ins[k - 1][1].split()[1])
                srcs = [int(v) for v in re.findall(r"\bv(\d+)\b", line.split(None, 2)[2])] if len(line.split(None, 2)) > 2 else []
                if tdst and int(tdst.group(1)) in srcs:
                    bad.append((kern, 0, ins[k - 1][1], line))
            reg = int(re.match(r"v(\d+)", line.split()[1]).group(1))
            wait = 0
            for kern2, nxt in ins[k + 1:k + 10]:
                if nxt.startswith("v_mfma"):
                    spans = re.findall(r"v\[(\d+):(\d+)\]", nxt)
                    if any(int(a) <= reg <= int(b) for a, b in spans[1:3]):   # (A and B operands; span 0 is the result)
                        checked += 1
                        if wait < 2:
                            bad.append((kern, wait, line, nxt))
                    break
                wait += int(nxt.split()[1]) + 1 if nxt.startswith("s_nop") else 1
    return checked, bad


if __name__ == "__main__":
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "torchani_amd", "libanihip.so")
    n, bad = check(path)
    for kern, wait, a, b in bad:
        print(f"{kern}: {wait} wait state(s) between\n    {a}\n    {b}")
    print(f"{n} inline-assembly VALU -> MFMA pairs checked, {len(bad)} with fewer than two wait states")
    sys.exit(1 if bad else 0)
