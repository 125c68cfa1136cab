"""Per-kernel resource usage (registers, scratch, spills, LDS) read from the code object inside a built libd3il_rollout*.so - no recompilation.
Usage: python tools/isa_resources.py [path/to/lib.so] [name-filter]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".so") else os.path.join(ROOT, "d3il_amd", "libd3il_rollout.so")
    flt = [a for a in sys.argv[1:] if not a.endswith(".so")]
    with tempfile.TemporaryDirectory() as td:
        co = os.path.join(td, "gfx950.co")
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--unbundle", "--input=" + lib,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], stderr=subprocess.DEVNULL) if False else None
        # the fat binary sits in the .hip_fatbin section of the host library
        fat = os.path.join(td, "fat.bin")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib])
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--unbundle", "--input=" + fat,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
        notes = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co]).decode()
    kernels = re.split(r"\n\s+- \.agpr_count:", notes)
    print("%-64s %5s %5s %5s %9s %9s %9s %8s" % ("kernel", "vgpr", "agpr", "sgpr", "scratch B", "sgpr spl", "vgpr spl", "LDS B"))
    for k in kernels[1:]:
        k = ".agpr_count:" + k
        def f(key):
            m = re.search(r"\.%s:\s+(\S+)" % key, k)
            return m.group(1) if m else "?"
        name = f("name")
        try:
            name = subprocess.check_output([os.path.join(LLVM, "llvm-cxxfilt"), name]).decode().strip()
        except Exception:
            pass
        name = re.sub(r"\(.*", "", name)
        if flt and not any(x in name for x in flt):
            continue
        print("%-64s %5s %5s %5s %9s %9s %9s %8s" % (name[:64], f("vgpr_count"), f("agpr_count"), f("sgpr_count"), f("private_segment_fixed_size"),
                                                   f("sgpr_spill_count"), f("vgpr_spill_count"), f("group_segment_fixed_size")))


if __name__ == "__main__":
    main()
