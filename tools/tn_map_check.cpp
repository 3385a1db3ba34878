// tn_map_check.cpp -- replays the index arithmetic of kronfluence_amd/csrc/kf_tn_map.h on the CPU, for each of its LDS images
// (g++ -std=c++17 -I kronfluence_amd/csrc tools/tn_map_check.cpp; tests/test_tools_cpu.py builds and runs it):
//   1. the 8 waves x 2 requests x 64 lanes of a piece write every 16-byte chunk of its 16 KB exactly once, at image(k, row);
//   2. under the lane semantics of ds_read_b64_tr_b16 (within a 16-lane group, lane l receives element l & 3 of the 64-bit words
//      addressed by lanes (l >> 2) + 4 j -- cdna_hip_programming.md T10, confirmed on the MI355X by tools/tr_probe.py), the
//      fragment a lane assembles for (block, k-slab) holds tile row block_row + (lane & 31) at k = 16 kk + 8 (lane >> 5) + 0..7 --
//      the operand layout of v_mfma_f32_32x32x16_bf16 -- for every wave, block and k-slab of both operands;
//   3. each 32-lane half of every transposing read touches all 64 LDS banks once (address arithmetic only: the hardware's own
//      conflict classes are what tools/tr_probe.py measures);
//   4. the (kk, quad) part of a word address is the same constant for every lane and block (kf_pingpong_tn.h folds it into the
//      instruction's immediate offset).
// Exit code 0 and "ok" when all hold.
#include <cstdint>
#include <cstdio>
#include <set>
#include <vector>

#include "kf_tn_map.h"

using namespace kf::tnmap;

template <int IMG>
int check() {
    using Img = Image<IMG>;
    int errors = 0;
    // the operand tile: value of (k, tile row f) = k * 256 + f
    for (int piece = 0; piece < 4; ++piece) {
        std::vector<int> lds(PIECE_BYTES / 2, -1);   // 16-bit elements of the piece's image
        for (int wave = 0; wave < 8; ++wave)
            for (int h = 0; h < 2; ++h)
                for (int lane = 0; lane < 64; ++lane) {
                    const int q = request_of(wave, h);
                    const int k = Img::dma_k(q, lane), fl = Img::dma_row(q, lane);
                    const int dst = q * REQUEST_BYTES + 16 * lane;   // lane-linear LDS-DMA write
                    if (k < 0 || k >= 64 || fl < 0 || fl + 8 > 128 || fl % 8 != 0) { ++errors; continue; }
                    if (dst != Img::at(k, fl)) { if (errors++ < 5) std::printf("image %d piece %d wave %d h %d lane %d: lands at %d, image says %d\n", IMG, piece, wave, h, lane, dst, Img::at(k, fl)); }
                    for (int e = 0; e < 8; ++e) {
                        if (lds[dst / 2 + e] != -1) { if (errors++ < 5) std::printf("image %d: chunk written twice at %d\n", IMG, dst); }
                        lds[dst / 2 + e] = k * 256 + tile_row(piece, fl + e);
                    }
                }
        for (int v : lds) if (v == -1) { if (errors++ < 5) std::printf("image %d piece %d: an element was never written\n", IMG, piece); break; }

        // fragments read from this piece
        struct Block { int fl0, tile_row0; };
        std::vector<Block> blocks;
        if (piece < 2) { for (int wm = 0; wm < 2; ++wm) for (int i = 2 * piece; i < 2 * piece + 2; ++i) if (a_piece(i) == piece) blocks.push_back({a_row(wm, i), wm * 128 + i * 32}); }
        else { for (int wn = 0; wn < 4; ++wn) if (b_piece(wn) == piece) for (int jn = 0; jn < 2; ++jn) blocks.push_back({b_row(wn, jn), wn * 64 + jn * 32}); }
        if (blocks.size() != 4) { ++errors; std::printf("piece %d: %zu blocks\n", piece, blocks.size()); }
        for (const Block& blk : blocks)
            for (int kk = 0; kk < 4; ++kk)
                for (int quad = 0; quad < 2; ++quad) {
                    int addr[64];
                    for (int lane = 0; lane < 64; ++lane) {
                        addr[lane] = word<IMG>(blk.fl0, kk, quad, lane);
                        const int step = word_step<IMG>(kk, quad);
                        if (addr[lane] - word<IMG>(blk.fl0, 0, 0, lane) != step) { if (errors++ < 5) std::printf("image %d: the (kk %d, quad %d) step is not a constant\n", IMG, kk, quad); }
                    }
                    for (int half = 0; half < 2; ++half) {   // banks of a 32-lane half: 2 dwords per lane
                        std::set<int> banks;
                        for (int lane = 32 * half; lane < 32 * half + 32; ++lane) { banks.insert((addr[lane] / 4) % 64); banks.insert((addr[lane] / 4 + 1) % 64); }
                        if (banks.size() != 64) { if (errors++ < 5) std::printf("image %d piece %d block %d kk %d quad %d half %d: %zu banks\n", IMG, piece, blk.fl0, kk, quad, half, banks.size()); }
                    }
                    for (int lane = 0; lane < 64; ++lane) {
                        if (addr[lane] % 8 != 0 || addr[lane] < 0 || addr[lane] + 8 > PIECE_BYTES) { ++errors; continue; }
                        const int group = lane & ~15, l = lane & 15;
                        for (int j = 0; j < 4; ++j) {
                            const int got = lds[addr[group + (l >> 2) + 4 * j] / 2 + (l & 3)];   // the transposing read
                            const int k = 16 * kk + 8 * (lane >> 5) + 4 * quad + j;
                            const int want = k * 256 + blk.tile_row0 + (lane & 31);
                            if (got != want) { if (errors++ < 8) std::printf("image %d piece %d block row %d kk %d quad %d lane %d j %d: got (k %d, row %d), want (k %d, row %d)\n", IMG, piece, blk.tile_row0, kk, quad, lane, j, got / 256, got % 256, want / 256, want % 256); }
                        }
                    }
                }
    }
    return errors;
}

int main() {
    const int errors = check<0>() + check<1>() + check<2>();
    std::printf(errors ? "%d errors\n" : "ok\n", errors);
    return errors ? 1 : 0;
}
