"""Device code objects of the built library: per-kernel resource notes and disassembly, without a GPU.
    python tools/code_objects.py            -> every kernel with VGPR spills / scratch (llvm-readelf --notes of each TU's gfx950 code object)
    python tools/code_objects.py --all      -> every kernel: vgpr / agpr / sgpr / spill / scratch / LDS
Used by tests/test_code_objects_cpu.py (no product kernel may spill; the M0 discipline of the inline-asm LDS-DMA statements)."""
import glob
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin/"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJ = os.path.join(ROOT, "omniserve_amd", "csrc", "build")
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"


def extract(obj, outdir):
    """gfx950 code object of one host object (the .hip_fatbin section, unbundled); returns its path (None: no device code)."""
    base = os.path.basename(obj)[:-2]
    fat, co = os.path.join(outdir, base + ".fatbin"), os.path.join(outdir, base + ".co")
    r = subprocess.run([LLVM + "llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, obj], capture_output=True, text=True)
    if r.returncode != 0:      # a host-only translation unit (the planner)
        return None
    subprocess.run([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat, "--targets=" + TARGET,
                    "--output=" + co], check=True, capture_output=True)
    return co


def kernels(co):
    """[{name, vgpr, agpr, sgpr, spill, scratch, lds}] from the code object's metadata notes."""
    notes = subprocess.run([LLVM + "llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
    out = []
    for blk in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
        g = lambda key: int(re.search(r"\.%s:\s+(\d+)" % key, blk).group(1))  # noqa: E731
        out.append(dict(name=re.search(r"\.name:\s+(\S+)", blk).group(1), agpr=int(blk.split()[0]), vgpr=g("vgpr_count"),
                        sgpr=g("sgpr_count"), spill=g("vgpr_spill_count"), sgpr_spill=g("sgpr_spill_count"),
                        scratch=g("private_segment_fixed_size"), lds=g("group_segment_fixed_size")))
    return out


def disassemble(co):
    return subprocess.run([LLVM + "llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True, check=True).stdout


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True)
    return r.stdout.split("\n")[:len(names)]


def all_objects():
    objs = sorted(glob.glob(os.path.join(OBJ, "*.o")))
    if not objs:
        raise RuntimeError("no objects under %s: run `python -m omniserve_amd.build` first" % OBJ)
    return objs


def main():
    show_all = "--all" in sys.argv
    with tempfile.TemporaryDirectory() as tmp:
        n = 0
        for obj in all_objects():
            co = extract(obj, tmp)
            if co is None:
                continue
            ks = kernels(co)
            names = demangle([k["name"] for k in ks])
            for k, d in zip(ks, names):
                if show_all or k["spill"] or k["scratch"]:
                    n += 1
                    print("%-18s vgpr %3d agpr %3d sgpr %3d spill %3d scratch %4d lds %6d  %s" % (
                        os.path.basename(obj), k["vgpr"], k["agpr"], k["sgpr"], k["spill"], k["scratch"], k["lds"], d[:140]))
        print("%d kernel(s) listed" % n)


if __name__ == "__main__":
    main()
