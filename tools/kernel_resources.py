"""Register / LDS / scratch budget of every kernel in the built library, read from the code objects' metadata (no GPU):

    python tools/kernel_resources.py > profiles/rNN_kernel_resources.txt

One row per kernel: VGPRs (arch + accumulation), SGPRs, static LDS, scratch bytes, VGPR / SGPR spills, max workgroup size.  Dynamic LDS (the
main loops take their 128 KB at launch) is not in the metadata; DESIGN.md section 4 lists it per kernel."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout
    return [re.sub(r"\(.*\)$", "", re.sub(r"\(anonymous namespace\)::|^void ", "", line)) for line in out.splitlines()]


def main():
    rows = []
    with tempfile.TemporaryDirectory() as work:
        for unit in ("kf_kernels", "kf_eigh", "kf_score_v2"):
            obj = os.path.join(work, unit + ".o")
            with open(os.path.join(ROOT, "kronfluence_amd", "csrc", "obj", unit + ".o"), "rb") as src, open(obj, "wb") as dst:
                dst.write(src.read())
            fat, code = os.path.join(work, unit + ".fat"), os.path.join(work, unit + ".co")
            subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", obj], check=True)
            subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={code}"], check=True)
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", code], capture_output=True, text=True, check=True).stdout
            for block in notes.split("\n  - .agpr_count:")[1:]:
                field = lambda key: re.search(rf"\.{key}:\s+(\S+)", block).group(1)   # noqa: E731
                rows.append((unit, field("name"), int(block.split()[0]), int(field("vgpr_count")), int(field("sgpr_count")),
                             int(field("group_segment_fixed_size")), int(field("private_segment_fixed_size")),
                             int(field("vgpr_spill_count")), int(field("sgpr_spill_count")), int(field("max_flat_workgroup_size"))))
    names = demangle([r[1] for r in rows])
    print(f"{'unit':12s} {'kernel':58s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'lds(static)':>11s} {'scratch':>8s} {'vspill':>6s} {'sspill':>6s} {'wg':>5s}")
    for (unit, _, agpr, vgpr, sgpr, lds, scratch, vspill, sspill, wg), name in sorted(zip(rows, names), key=lambda x: (x[0][0], x[1])):
        print(f"{unit:12s} {name[:58]:58s} {vgpr:5d} {agpr:5d} {sgpr:5d} {lds:11d} {scratch:8d} {vspill:6d} {sspill:6d} {wg:5d}")
    bad = [n for r, n in zip(rows, names) if r[6] or r[7]]
    soft = [n for r, n in zip(rows, names) if r[8]]
    print(f"\n{len(rows)} kernels; with scratch memory or VGPR spills: {bad if bad else 'none'}")
    print(f"SGPR spills (to VGPR lanes, no memory): {len(soft)} kernels, at most {max([r[8] for r in rows])} registers")


if __name__ == "__main__":
    sys.exit(main())
