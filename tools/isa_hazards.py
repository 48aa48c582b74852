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


def _regs(tok):
    """VGPR numbers named by one operand token ('v7', 'v[4:7]', '-v3', '|v2|'); empty for anything else."""
    m = re.fullmatch(r"[-|]*v(\d+)\|?", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"[-|]*v\[(\d+):(\d+)\]\|?", tok)
    return set(range(int(m.group(1)), int(m.group(2)) + 1)) if m else set()


def _operands(line):
    parts = line.split(None, 1)
    if len(parts) < 2:
        return []
    body = re.split(r"\s+(?:op_sel|op_sel_hi|neg_lo|neg_hi|quad_perm|row_|bank_mask|bound_ctrl|offset|clamp|mul:|div:|cbsz|abid|blgp|sc0|sc1|nt|fi:)", parts[1])[0]
    return [t.strip() for t in body.split(",")]


TRANS = re.compile(r"v_(exp|log|rcp|rsq|sqrt|sin|cos)_")
VALU = re.compile(r"v_(?!mfma|readlane|readfirstlane|nop)")
DPP = re.compile(r"(quad_perm|row_shl|row_shr|row_ror|row_mirror|row_half_mirror|row_bcast|wave_shl|wave_shr|wave_rol|wave_ror|row_newbcast|row_share|row_xmask)")


def _states(nxt):
    return int(nxt.split()[1]) + 1 if nxt.startswith("s_nop") else 1


def check(lib_path):
    """(pairs checked, violations).  Walks EVERY instruction of every gfx950 kernel of the library -- compiler-scheduled
    code and the inline-assembly sites (csrc/mlp.hip: v_fma_mix{lo,hi}_f16, s_nop) alike -- for the data hazards the hardware
    does not interlock (gfx950 ISA guide / LLVM's GCNHazardRecognizer), by the number of wait states between producer and
    consumer (an instruction in between = 1, s_nop N = N + 1):
      (a) VALU write of a VGPR -> MFMA reading it as A / B operand: 2      [round 3's k_gemm_l0b bug: 1 behind inline asm]
      (b) transcendental (v_exp / log / rcp / rsq / sqrt / sin / cos) result -> non-transcendental VALU reading it: 1
      (c) VALU write of a VGPR -> DPP instruction reading it: 2
      (d) VALU write of a VGPR -> v_readlane / v_readfirstlane of it: 1
      (e) VALU write of a VGPR -> v_permlane{16,32}_swap using it: 2
      (f) v_readlane / v_readfirstlane write of an SGPR -> vector-memory instruction using it as (part of) its scalar base: 5
          [round 6: the fused kernel's inline-assembly queue draw faulted behind the v_readlane that restores a spilled pointer]
    hipcc pads these for the code it schedules itself; behind asm() it cannot see the producer."""
    checked, bad = 0, []
    for code in code_objects(lib_path):
        c, b = check_instructions(list(instructions(code)))
        checked += c
        bad += b
    return checked, bad


def check_instructions(ins):
    """The walk of ``check`` over a list of (kernel name, instruction text)."""
    checked, bad = 0, []
    if True:
        for k, (kern, line) in enumerate(ins):
            if not VALU.match(line) or line.startswith("v_cmp"):
                continue
            ops = _operands(line)
            dst = _regs(ops[0]) if ops else set()
            if not dst:
                continue
            is_trans = bool(TRANS.match(line))
            wait = 0
            for kern2, nxt in ins[k + 1:k + 4]:
                if kern2 != kern or nxt.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
                    break
                if nxt.startswith("s_nop"):
                    wait += _states(nxt)
                    continue
                nops = _operands(nxt)
                srcs = set().union(*[_regs(t) for t in nops[1:]]) if len(nops) > 1 else set()
                need, what = 0, None
                if nxt.startswith("v_mfma"):
                    ab = set().union(*[_regs(t) for t in nops[1:3]])
                    if dst & ab:
                        need, what = 2, "VALU -> MFMA A/B"
                elif nxt.startswith(("v_readlane", "v_readfirstlane")):
                    if dst & srcs:
                        need, what = 1, "VALU -> readlane"
                elif nxt.startswith("v_permlane") and "swap" in nxt:
                    if dst & set().union(*[_regs(t) for t in nops]):
                        need, what = 2, "VALU -> permlane swap"
                elif DPP.search(nxt) and nxt.startswith("v_"):
                    if dst & srcs:
                        need, what = 2, "VALU -> DPP"
                elif is_trans and VALU.match(nxt) and not TRANS.match(nxt):
                    if dst & srcs:
                        need, what = 1, "trans -> VALU"
                if what:
                    checked += 1
                    if wait < need:
                        bad.append((kern, wait, what + ": " + line, nxt))
                # a consumer that overwrites the register ends the producer's reach
                if nops and (_regs(nops[0]) & dst) and not nxt.startswith(("v_mfma",)):
                    break
                wait += 1
    # (f) VALU write of an SGPR -> VMEM reading it as scalar base
    for k, (kern, line) in enumerate(ins):
        m = re.match(r"v_(?:readlane|readfirstlane)_b32 s(\d+),", line)
        if not m:
            continue
        sreg, wait = int(m.group(1)), 0
        for kern2, nxt in ins[k + 1:k + 7]:
            if kern2 != kern or nxt.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
                break
            if nxt.startswith("s_nop"):
                wait += _states(nxt)
                continue
            if nxt.startswith(("global_", "buffer_", "scratch_", "flat_")):
                used = set()
                for a, b in re.findall(r"\bs\[(\d+):(\d+)\]", nxt):
                    used |= set(range(int(a), int(b) + 1))
                used |= {int(a) for a in re.findall(r"\bs(\d+)\b", nxt)}
                if sreg in used:
                    checked += 1
                    if wait < 5:
                        bad.append((kern, wait, "readlane SGPR -> VMEM base: " + line, nxt))
                    break
            if re.match(r"(?:s_\w+|v_(?:readlane|readfirstlane)_b32) s(?:\[%d:\d+\]|%d)\b" % (sreg, sreg), nxt):
                break   # (the register is written again)
            wait += 1
    return checked, bad


if __name__ == "__main__":
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "torchani_amd", "libanihip.so")
    n, bad = check(path)
    for kern, wait, a, b in bad:
        print(f"{kern}: {wait} wait state(s) between\n    {a}\n    {b}")
    print(f"{n} producer -> consumer pairs checked (VALU -> MFMA / DPP / readlane / permlane-swap, trans -> VALU), "
          f"{len(bad)} with too few wait states")
    sys.exit(1 if bad else 0)
