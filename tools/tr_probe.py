"""Driver of tools/tr_probe.hip (lane semantics and LDS-array cycles of gfx950's transposing LDS read).

    hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/tr_probe.hip -o tools/libtr_probe.so     (here: cross-compiles)
    gpurun -- 'python tools/tr_probe.py'

Part 1 (semantics): for a few per-lane address patterns prints, per lane, which LDS positions (16-bit units) it received, and checks
the rule the design of a TN operand path assumes (cdna_hip_programming.md T10: within a 16-lane group, lane l receives element l & 3 of
the 64-bit words of lanes (l >> 2) + 4 j, j = 0..3 -- a [4][16] block read column-wise when lane s addresses row s >> 2, columns
4 (s & 3) ..).  Part 2 (cycles): LDS-array cycles per read for candidate LDS images of a 32 x 16 MFMA operand block fetched from a
K-major (``[t][feature]``) tile, next to plain ds_read_b64 / ds_read_b128 on conflict-free addresses as the yardstick.
"""
import ctypes
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
DEV = "cuda:0"


def lib():
    handle = ctypes.CDLL(os.path.join(HERE, "libtr_probe.so"))
    handle.tr_probe_semantics.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    handle.tr_probe_cycles.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                       ctypes.c_void_p, ctypes.c_void_p]
    return handle


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def semantics(handle, addr):
    a = torch.tensor(addr, dtype=torch.int32, device=DEV)
    out = torch.zeros(64 * 4, dtype=torch.int16, device=DEV)
    rc = handle.tr_probe_semantics(a.data_ptr(), out.data_ptr(), stream())
    torch.cuda.synchronize()
    assert rc == 0, rc
    return (out.cpu().to(torch.int32) & 0xFFFF).reshape(64, 4).tolist()


def expected(addr):
    """The assumed rule: lane l of a 16-lane group, element j <- element (l & 3) of the word addressed by lane (l >> 2) + 4 j."""
    out = []
    for lane in range(64):
        group, l = lane & ~15, lane & 15
        out.append([addr[group + (l >> 2) + 4 * j] // 2 + (l & 3) for j in range(4)])
    return out


def operand_addresses(layout, o0, k0, second):
    """Per-lane byte addresses of ONE transposing read of the 32 (rows o) x 16 (k) MFMA operand block at (o0, k0): lanes 0-31 hold
    k0 .. k0 + 7, lanes 32-63 k0 + 8 .. 15; a lane's two reads deliver k-quads 0 and 1 of its eight (second = 1: the latter)."""
    addr = []
    for lane in range(64):
        group, s = lane >> 4, lane & 15
        t = k0 + 8 * (group >> 1) + 4 * second + (s >> 2)
        o = o0 + 16 * (group & 1) + 4 * (s & 3)
        addr.append(layout(t, o))
    return addr


# candidate LDS images of a 64 (t) x 256 (o) bf16 tile; every 16-byte chunk (8 o of one t) stays contiguous (LDS-DMA granularity)
def plain(t, o):                # [t][o], 512-byte rows
    return t * 512 + o * 2


def rows128(t, o):              # [o / 64][t][64 o]: 128-byte rows, 8 KB per 64-column panel
    return (o >> 6) * 8192 + t * 128 + (o & 63) * 2


def rows128_half_swizzle(t, o):  # as rows128, the two 64-byte halves of a row swapped on rows with bit 1 of t set
    half = ((o >> 5) & 1) ^ ((t >> 1) & 1)
    return (o >> 6) * 8192 + t * 128 + half * 64 + (o & 31) * 2


def rows64(t, o):               # [o / 32][t][32 o]: 64-byte rows (four t rows = one 256-byte bank row)
    return (o >> 5) * 4096 + t * 64 + (o & 31) * 2


def rows64_swapped(t, o):       # as rows64 with bits 2 and 3 of t swapped in the row index (kf_tn_map.h IMG 2; the guide's [8-key][32-col] subtile)
    row = (t & ~12) | ((t & 4) << 1) | ((t & 8) >> 1)
    return (o >> 5) * 4096 + row * 64 + (o & 31) * 2


def rows32(t, o):               # [o / 16][t][16 o]: 32-byte rows -- the guide's [k][16-col] subtile
    return (o >> 4) * 2048 + t * 32 + (o & 15) * 2


def rows256(t, o):              # [o / 128][t][128 o]: 256-byte rows = what an LDS-DMA request of 4 t rows writes lane-linearly
    return (o >> 7) * 16384 + t * 256 + (o & 127) * 2


def rows256_quarter_swizzle(t, o):  # the planned image: the 64-byte quarter of a row XORed with t & 3
    quarter = ((o >> 5) & 3) ^ (t & 3)
    return (o >> 7) * 16384 + t * 256 + quarter * 64 + (o & 31) * 2


LAYOUTS = [("256-B rows", rows256), ("256-B rows, quarters swizzled by t & 3 (IMG 0)", rows256_quarter_swizzle), ("plain [t][256 o]", plain), ("128-B rows", rows128), ("128-B rows, halves swizzled by t bit 1", rows128_half_swizzle),
           ("64-B rows (IMG 1)", rows64), ("64-B rows, t bits 2 / 3 swapped (IMG 2)", rows64_swapped), ("32-B rows ([k][16] subtiles)", rows32)]


def main():
    handle = lib()
    print("== semantics (positions in 16-bit units; a lane's four values)")
    patterns = [("linear, lane * 8 bytes", [lane * 8 for lane in range(64)]),
                ("every lane the same address 64", [64] * 64),
                ("[4 t][16 o] blocks of a plain [t][256 o] image, rows 512 B", operand_addresses(plain, 0, 0, 0)),
                ("reversed lanes", [(63 - lane) * 8 for lane in range(64)])]
    rule_holds = True
    for name, addr in patterns:
        got, want = semantics(handle, addr), expected(addr)
        ok = got == want
        rule_holds &= ok
        print(f"  {name}: assumed rule {'HOLDS' if ok else 'DOES NOT HOLD'}")
        if not ok:
            for lane in (0, 1, 2, 3, 4, 15, 16, 17, 32, 63):
                print(f"    lane {lane:2d}: address {addr[lane]:5d} got {got[lane]} assumed {want[lane]}")
    print("== LDS-array cycles per read (8 waves; the k-th of 8 unrolled reads at + k * step bytes)")
    iters = 2000
    tables, names, steps = [], [], []
    for name, layout in LAYOUTS:
        for second in (0, 1):
            tables.append(operand_addresses(layout, 0, 0, second))
            names.append(f"{name}, k-quad {second}")
    sink = torch.zeros(1, dtype=torch.int32, device=DEV)

    def cycles_of(table, step, mode):
        rows = torch.tensor([table], dtype=torch.int32, device=DEV)
        cycles = torch.zeros(1, dtype=torch.int64, device=DEV)
        rc = handle.tr_probe_cycles(rows.data_ptr(), 1, iters, step, mode, cycles.data_ptr(), sink.data_ptr(), stream())
        torch.cuda.synchronize()
        assert rc == 0, rc
        return float(cycles[0]) / (iters * 8 * 8)   # 8 waves x 8 unrolled reads share the LDS array

    for index, name in enumerate(names):
        layout = LAYOUTS[index // 2][1]
        walk = layout(0, 32) - layout(0, 0)   # the next 32 rows (o) of the operand: what consecutive fragment reads of a wave do
        print(f"  ds_read_b64_tr_b16  {name:62s} same block {cycles_of(tables[index], 0, 0):6.2f}   walking o by 32 "
              f"{cycles_of(tables[index], walk, 0):6.2f} cycles per wave-read")
    # yardsticks: conflict-free plain reads
    for mode, label, table in ((1, "ds_read_b64 (lane * 8)", [lane * 8 for lane in range(64)]),
                               (2, "ds_read_b128 (lane * 16)", [lane * 16 for lane in range(64)]),
                               (0, "ds_read_b64_tr_b16 (lane * 8)", [lane * 8 for lane in range(64)]),
                               (1, "ds_read_b64, 32-way conflict (lane * 256)", [(lane * 256) % 65536 for lane in range(64)])):
        print(f"  yardstick {label:46s} {cycles_of(table, 1024 if 'conflict' not in label else 8, mode):6.2f} cycles per wave-read")
    return 0 if rule_holds else 1


if __name__ == "__main__":
    sys.exit(main())
