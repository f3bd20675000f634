// Host-side check of the chunk tables of csrc/jk_mx.hip (MxGeom, jk_m4_layout.h) for every size the kernel is instantiated
// for: the chunks tile the triangle of 4 x 4 blocks exactly and in order, each fits a ring buffer, the workgroup's LDS stays
// within 160 KB, block -> (chunk, offset) -> block round-trips for every block, and band segments are whole column groups of
// one band.  Built and run by tests/test_abi.py::test_jk_mx_chunk_tables (no GPU needed: nothing is launched).
#include <cstdio>

#include "jk_m4_layout.h"

template <int NB>
int check() {
    using G = MxGeom<NB>;
    int bad = 0, tot = 0;
    for (int k = 0; k < G::NCH; ++k) {
        const MxChunk c = G::CH.c[k];
        if (c.start != tot) ++bad;
        tot += c.blocks;
        if (c.blocks <= 0 || c.blocks > 32 * G::LPTM || G::lpt(k) > 6 || G::lpt(k) < 1) ++bad;
        if (c.band) {
            if (c.rb != c.ra + 4 || c.ra % 4 != 0 || c.j0 < 0 || c.j1 <= c.j0 || c.j1 > c.ra / 4 + 1) ++bad;
            if (c.blocks != mx_band_blocks(c.ra / 4, c.j0, c.j1)) ++bad;
        } else {
            if (c.blocks != m4_tri(c.rb) - m4_tri(c.ra)) ++bad;
        }
    }
    if (tot != G::NBLK || !G::CH.c[0].band == false) ++bad;  // (chunk 0 is a whole-row chunk: the walk loads X there)
    if ((MX_RING * G::BUF + G::FIXED) * 8 > MX_LDS_BYTES) ++bad;
    for (int T = 0; T < NB; ++T)
        for (int C = 0; C <= T; ++C) {
            int k, off, T2, C2;
            mx_locate(G::CH, T, C, k, off);
            if (k < 0 || off < 0 || off >= G::CH.c[k].blocks) {
                ++bad;
                continue;
            }
            mx_block_of(G::CH, k, off, T2, C2);
            if (T2 != T || C2 != C) ++bad;
        }
    std::printf("NB=%d N=%d NCH=%d LPTM=%d lds=%d bad=%d\n", NB, G::N, G::NCH, G::LPTM, (MX_RING * G::BUF + G::FIXED) * 8, bad);
    return bad;
}

#define X(NB_) bad += check<NB_>();
int main() {
    int bad = 0;
    X(38) X(40) X(42) X(44) X(46) X(48) X(50) X(52) X(54) X(56) X(58) X(60) X(62) X(64)
    X(68) X(72) X(76) X(80) X(84) X(88) X(92) X(96) X(100)
    return bad ? 1 : 0;
}
