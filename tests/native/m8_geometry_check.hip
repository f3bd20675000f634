// Host-side check of the geometry of csrc/jk_m8.hip (M8Geom, jk_m8_layout.h) for the instantiated size and several chunk
// sizes: the chunks tile the triangle of 4 x 4 blocks in whole block rows, each fits a ring buffer, ring and LDS budgets,
// nk(p) = the chunk that holds block row p / 4, tile addresses = running sum of the tile lengths, the staging index is a
// bijection of the lower-triangle elements into their chunk's slots and -- with at most four chunks -- equals jk_m4.hip's index
// formula with the chunk boundaries nbx_jk_m8_weight_layout hands to the scalars kernel.
// Built and run by tests/test_abi.py::test_jk_m8_geometry (no GPU needed: nothing is launched).
#include <cstdio>
#include <vector>

#include "jk_m8_layout.h"

template <int NB, int LP>
int check() {
    using G = M8Geom<NB, LP>;
    int bad = 0, tot = 0;
    for (int k = 0; k < G::NCH; ++k) {
        if (G::start(k) != tot) ++bad;
        if (G::blocks(k) <= 0 || G::blocks(k) > G::CAP) ++bad;
        if (G::row0(k + 1) <= G::row0(k)) ++bad;
        tot += G::blocks(k);
    }
    if (tot != G::NBLK || G::row0(0) != 0 || G::row0(G::NCH) != NB) ++bad;
    if ((G::RING * G::BUF + G::FIXED) * 8 > M8_LDS_BYTES || G::RING < 4 || (G::RING - 2) * LP > 63) ++bad;
    int64_t off = 0;
    int prev_nk = 1;
    for (int p = 0; p < G::N; ++p) {
        const int nk = m8_nk<G>(p);
        if (nk < prev_nk || nk < 1 || nk > G::NCH) ++bad;
        prev_nk = nk;
        if (!(G::row0(nk - 1) <= (p >> 2) && (p >> 2) < G::row0(nk))) ++bad;
        if (m8_len<G>(p) != 16 * m4_tri(G::row0(nk))) ++bad;
        for (int q = 0; q <= p; ++q) {
            if (m8_tile_offset<G>((int64_t)p * (p + 1) / 2 + q) != off) ++bad;
            off += m8_len<G>(p);
        }
    }
    if (m8_tile_offset<G>(m4_tri(G::N)) != off) ++bad;
    std::vector<char> seen((size_t)G::NCH * LP * M4_PROD_THREADS * 2, 0);
    for (int r = 0; r < G::N; ++r)
        for (int c = 0; c <= r; ++c) {
            const int i = m8_stage_index<G>(r, c);
            if (i < 0 || i >= (int)seen.size() || seen[i]) {
                ++bad;
                continue;
            }
            seen[i] = 1;
            const int k = i / (LP * M4_PROD_THREADS * 2), d = i % (LP * M4_PROD_THREADS * 2);
            if (d >= 16 * G::blocks(k)) ++bad;
            if (G::NCH <= 4 && i != m4_weight_index_rt(G::row0(1), G::row0(2), G::row0(3), LP, r, c)) ++bad;
        }
    std::printf("NB=%d LP=%d NCH=%d RING=%d lds=%d packed8=%lld bad=%d\n", NB, LP, G::NCH, G::RING, (G::RING * G::BUF + G::FIXED) * 8,
                (long long)off * 8, bad);
    return bad;
}

int main() {
    int bad = 0;
    bad += check<37, 3>();
    bad += check<37, 4>();
    bad += check<37, 5>();
    // the instances (jk_m8.hip NBX_M8_SIZES at their own chunk size)
    bad += check<25, m8_lp(25)>();
    bad += check<26, m8_lp(26)>();
    bad += check<27, m8_lp(27)>();
    bad += check<28, m8_lp(28)>();
    bad += check<29, m8_lp(29)>();
    bad += check<30, m8_lp(30)>();
    bad += check<31, m8_lp(31)>();
    bad += check<32, m8_lp(32)>();
    bad += check<33, m8_lp(33)>();
    bad += check<34, m8_lp(34)>();
    bad += check<35, m8_lp(35)>();
    bad += check<36, m8_lp(36)>();
    bad += check<37, m8_lp(37)>();
    return bad ? 1 : 0;
}
