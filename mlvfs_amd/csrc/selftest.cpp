// selftest.cpp -- checks that need no GPU: the selection networks of
// median_nets.h (compiled for the host here) against std::nth_element, the strip
// composition used by k_frame.hip, and the 16-bit table re-encodings.
#include "common.h"

#include <algorithm>
#include <cstdlib>
#include <vector>

#define MLV_NET_FN static inline
#define mlv_mn(a, b) ((a) < (b) ? (a) : (b))
#define mlv_mx(a, b) ((a) > (b) ? (a) : (b))
#include "median_nets.h"

namespace {

uint32_t rng_state = 12345;
uint32_t rng() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 17; rng_state ^= rng_state << 5; return rng_state; }

int value(int mode)
{
    switch (mode) {
        case 0: return (int)(rng() % 7) - 3;                 // many ties
        case 1: return (int)rng();                           // full int range incl. wrap-like values
        case 2: return (rng() & 1) ? INT32_MIN : (int)(rng() % 100);
        default: return (int)(rng() % 2000000) - 1000000;
    }
}

void sort5_host(int (&v)[5]) { std::sort(v, v + 5); }

// the composition of k_frame.hip's strip_median25 on host data
void strip25(const int plane[5][12], int (&med)[8])
{
    int col[12][5];
    for (int c = 0; c < 12; c++) { for (int r = 0; r < 5; r++) col[c][r] = plane[r][c]; sort5_host(col[c]); }
    int pr[6][10], qd[5][6];
    for (int p = 0; p < 6; p++) mlv_merge55(col[2 * p], col[2 * p + 1], pr[p]);
    for (int q = 0; q < 5; q++) mlv_quad_mid6(pr[q], pr[q + 1], qd[q]);
    for (int c = 0; c < 8; c++) {
        const int x = c + 2;
        int o[1];
        if (x % 2 == 0) mlv_final6of11(qd[(x - 2) / 2], col[x + 2], o);
        else mlv_final6of11(qd[(x - 1) / 2], col[x - 2], o);
        med[c] = o[0];
    }
}

}  // namespace

extern "C" int mlvfs_amd_selftest_host(void)
{
    int fails = 0;
    for (int mode = 0; mode < 4; mode++) {
        for (int it = 0; it < 20000; it++) {
            int v25[25], v9[9], v5[5], o[1];
            for (int &x : v25) x = value(mode);
            for (int i = 0; i < 9; i++) v9[i] = v25[i];
            for (int i = 0; i < 5; i++) v5[i] = v25[i];
            std::vector<int> s(v25, v25 + 25);
            std::sort(s.begin(), s.end());
            mlv_median25(v25, o); if (o[0] != s[12]) fails++;
            s.assign(v9, v9 + 9); std::sort(s.begin(), s.end());
            mlv_median9(v9, o); if (o[0] != s[4]) fails++;
            s.assign(v5, v5 + 5); std::sort(s.begin(), s.end());
            mlv_median5(v5, o); if (o[0] != s[2]) fails++;
        }
        for (int it = 0; it < 5000; it++) {
            int plane[5][12], med[8];
            for (auto &row : plane) for (int &x : row) x = value(mode);
            strip25(plane, med);
            for (int c = 0; c < 8; c++) {
                std::vector<int> s;
                for (int r = 0; r < 5; r++) for (int k = 0; k < 5; k++) s.push_back(plane[r][c + k]);
                std::sort(s.begin(), s.end());
                if (med[c] != s[12]) fails++;
            }
        }
    }
    if (!mlv::luts_ok()) fails += 1000000;
    return fails;
}

// The library's host EV tables (its own formulas, or the caller's get_raw2ev / get_ev2raw when the caller exports them) against
// tables handed in: raw2ev_lin[16384] (index = pixel - black; [0] = INT_MIN) and ev2raw[24 * 32768] (index 0 = EV -10 * 32768).
// 0 = identical.  Needs no GPU.  (The CPU test suite passes the tables of the reference's own main.c.)
extern "C" int mlvfs_amd_selftest_tables(const int32_t *raw2ev_lin, const int32_t *ev2raw)
{
    if (!mlv::luts_ok()) return 1;
    const int32_t *a = mlv::host_raw2ev_lin(), *b = mlv::host_ev2raw();
    for (int i = 0; i < 16384; i++) if (a[i] != raw2ev_lin[i]) return 2;
    for (int i = 0; i < 24 * MLV_EV_RES; i++) if (b[i] != ev2raw[i]) return 3;
    return 0;
}
