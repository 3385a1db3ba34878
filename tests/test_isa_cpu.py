"""What the compiler made of the wave-role-split main loops, read off the built gfx950 code object (no GPU needed).

The 256 x 256 loops (kf_pingpong.h, kf_pingpong_tn.h) are only as fast as their waits are COUNTED: eight LDS-DMA requests per
k-tile stay in flight across the MFMA groups and a wave waits for "all but N", never for all.  Twice this broke silently at the
compiler's hands, not in the source: hipcc puts ``s_waitcnt vmcnt(0)`` in front of the ``ds_read_tr16_b64`` builtin and in front of
plain LDS stores while LDS-DMA is outstanding (round 5: hence the inline-asm reads of kf_pingpong_tn.h).  The kernels stay correct
when that happens -- only slower -- so no parity test notices.  This one disassembles the object ``__graft_entry__.build()`` left in
``csrc/obj`` and checks, per kernel on those loops, the innermost loop that holds one k-tile's 32 MFMAs: 8 ``global_load_lds_dwordx4``
requests, exactly the two counted waits of its request schedule (ISSUE 0: 8 / 6, 1: 6 / 2, 2: 6 / 4) and no ``vmcnt(0)``; the K-major
loops in addition: 48 transposing ``ds_read_b64_tr_b16`` fragment reads and no other LDS read of fragments."""

import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJ = os.path.join(ROOT, "kronfluence_amd", "csrc", "obj", "kf_score_v2.o")
LLVM = "/opt/rocm/lib/llvm/bin"
WAITS = {0: ["vmcnt(8)", "vmcnt(6)"], 1: ["vmcnt(6)", "vmcnt(2)"], 2: ["vmcnt(6)", "vmcnt(4)"]}


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    tools = [os.path.join(LLVM, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump")]
    if not os.path.exists(OBJ) or not all(os.path.exists(t) for t in tools):
        pytest.skip("needs the object files of __graft_entry__.build() and the ROCm LLVM tools")
    work = tmp_path_factory.mktemp("isa")
    fat, code = str(work / "fat.bin"), str(work / "device.co")
    local = str(work / "kf_score_v2.o")
    shutil.copy(OBJ, local)   # llvm-objcopy rewrites its input when asked to dump a section
    subprocess.run([tools[0], "--dump-section", f".hip_fatbin={fat}", local], check=True)
    subprocess.run([tools[1], "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={code}"],
                   check=True)
    text = subprocess.run([tools[2], "-d", "--no-show-raw-insn", code], check=True, capture_output=True, text=True).stdout.splitlines()
    heads = [i for i, line in enumerate(text) if re.match(r"^[0-9a-f]+ <", line)]
    out = {}
    for n, i in enumerate(heads):
        name = re.search(r"<(.*)>:", text[i]).group(1)
        body = []
        for line in text[i + 1:(heads[n + 1] if n + 1 < len(heads) else len(text))]:
            m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):", line)
            if m:
                target = re.search(r"\+0x([0-9a-f]+)>", line)
                body.append((int(m.group(3), 16), m.group(1), m.group(2), int(target.group(1), 16) if target else None))
        if body:
            out[name] = body
    return out


def k_tile_loop(body):
    """The smallest loop (backward branch .. its target) that contains MFMAs."""
    base, best = body[0][0], None
    for address, op, _args, target in body:
        if (op.startswith("s_cbranch") or op == "s_branch") and target is not None and base + target < address:
            region = [x for x in body if base + target <= x[0] <= address]
            if any(x[1].startswith("v_mfma") for x in region) and (best is None or len(region) < len(best)):
                best = region
    return best


def template_ints(name):
    return [int(x) for x in re.findall(r"Li(\d+)E", name)]


def test_role_split_loops_wait_on_counts_not_on_everything(kernels):
    checked = 0
    for name, body in kernels.items():
        m = re.search(r"(score_gemm_v3|rotate_gemm_v3|psg_gemm_pp|cov_gemm_v3|psg_gemm_tn|cov_gemm_tn)_kernel", name)
        if not m:
            continue
        k_major = m.group(1).endswith("_tn")
        issue = 1 if k_major else template_ints(name)[-1]   # the K-major loops have one schedule; the others carry ISSUE last
        loop = k_tile_loop(body)
        assert loop is not None, name
        ops = [x[1] for x in loop]
        assert sum(op.startswith("v_mfma_f32_32x32x16_bf16") for op in ops) == 32, name
        assert sum(op.startswith("global_load_lds_dwordx4") for op in ops) == 8, name
        waits = [x[2] for x in loop if x[1] == "s_waitcnt" and "vmcnt" in x[2]]
        assert sorted(waits) == sorted(WAITS[issue]), (name, waits)   # (the compiler may rotate the loop: same two waits either way)
        assert ops.count("s_barrier") >= 2, name
        if k_major:
            assert ops.count("ds_read_b64_tr_b16") == 48, name
            assert not any(op.startswith("ds_read_b128") or op.startswith("ds_read2") for op in ops), name
        checked += 1
    # score (3 schedules), rotations (2 x 3), per-sample gradients (3), covariance (3), K-major gradients / covariance (3 images each)
    assert checked == 3 + 6 + 3 + 3 + 3 + 3, checked


def segments_of(loop):
    """The k-tile loop cut at its raw barriers: per segment (fragment reads, MFMAs, LDS-DMA requests, counted vm waits)."""
    segs = [[]]
    for x in loop:
        if x[1] == "s_barrier":
            segs.append([])
        else:
            segs[-1].append(x)
    out = []
    for seg in segs:
        ops = [x[1] for x in seg]
        out.append((sum(op.startswith("ds_read") for op in ops), sum(op.startswith("v_mfma") for op in ops),
                    sum(op.startswith("global_load_lds") for op in ops),
                    [int(re.search(r"vmcnt\((\d+)\)", x[2]).group(1)) for x in seg if x[1] == "s_waitcnt" and "vmcnt" in x[2]]))
    # the loop's first and last pieces belong to one segment (the back edge sits inside it)
    first, last = out[0], out[-1]
    merged = (first[0] + last[0], first[1] + last[1], first[2] + last[2], last[3] + first[3])
    return [merged] + out[1:-1]


def test_compiled_segments_are_the_ones_the_interval_model_checks(kernels):
    """tools/pp_schedule_check.py proves the RAW / WAR ordering of a request schedule that was transcribed BY HAND from
    kf_pingpong.h; this ties the transcription to the compiled code: for every kernel on the 256 x 256 loops the k-tile loop, cut at its
    four barriers, must be L M L M with -- segment by segment -- the model's number of requests (two DMA instructions per piece and
    wave) and the model's counted wait, fragment reads only in the L segments and 16 MFMAs in each M segment."""
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pp_schedule_check as model

    checked = 0
    for name, body in kernels.items():
        m = re.search(r"(score_gemm_v3|rotate_gemm_v3|psg_gemm_pp|cov_gemm_v3|psg_gemm_tn|cov_gemm_tn)_kernel", name)
        if not m:
            continue
        issue = 1 if m.group(1).endswith("_tn") else template_ints(name)[-1]
        program = model.program(issue, 8)
        steady = program[1 + 4 * 3: 1 + 4 * 4]          # the four segments of k-tile 3: both look-aheads exist
        want = [(kind, 2 * len(issues), wait) for kind, _reads, issues, wait in steady]
        segs = segments_of(k_tile_loop(body))
        assert len(segs) == 4, (name, len(segs))
        got = [("L" if reads and not mfma else "M" if mfma == 16 and not reads else "?", dma, waits[0] if waits else None)
               for reads, mfma, dma, waits in segs]
        assert all(len(waits) <= 1 for _, _, _, waits in segs), name
        rotations = [got[k:] + got[:k] for k in range(4)]   # the compiler may start the loop at any barrier
        assert want in rotations, (name, got, want)
        checked += 1
    assert checked == 21, checked
