"""Register / spill / LDS usage of the kernels inside libanihip.so (from the code objects' metadata notes).

    python tools/kernel_regs.py [path/to/libanihip.so] [name filter]
"""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from isa_hazards import LLVM, code_objects  # noqa: E402


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".so") else os.path.join(
        os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "torchani_amd", "libanihip.so")
    filt = [a for a in sys.argv[1:] if not a.endswith(".so")]
    for code in code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(code)
            f.flush()
            text = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", f.name], check=True,
                                  capture_output=True, text=True).stdout
        for blk in text.split("  - .agpr_count:")[1:]:
            get = lambda k: (re.search(r"\." + k + r":\s*(\S+)", blk) or [None, "?"])[1]
            name = get("name")
            try:
                name = subprocess.run([os.path.join(LLVM, "llvm-cxxfilt"), name], capture_output=True, text=True).stdout.strip()
            except OSError:
                pass
            name = re.sub(r"\(.*", "", name).replace("void ", "")
            if filt and not any(x in name for x in filt):
                continue
            print(f"{name:48s} vgpr {get('vgpr_count'):>4s} agpr {blk.split()[0]:>3s} spill {get('vgpr_spill_count'):>3s} "
                  f"sgpr {get('sgpr_count'):>4s} lds {get('group_segment_fixed_size'):>6s} scratch {get('private_segment_fixed_size'):>5s}")


if __name__ == "__main__":
    main()
