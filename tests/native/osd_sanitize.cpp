// osd_sanitize.cpp -- the BP+OSD host step (ldpcdecoders.jl_amd/csrc/osd_host.cpp, ldpc_osd_*) under
// AddressSanitizer + UndefinedBehaviorSanitizer, checked against the dense literal oracle
// (oracle/osd_oracle.c) on random and rank-deficient matrices, ties in the reliability key, all OSD
// orders the reference tests use, one and several host threads.  CPU build only (the GPU pool offers no
// sanitizers); built and run by tests/test_osd_host.py::test_osd_host_under_sanitizers.
// Exit code 0 = every estimate identical; prints the first mismatch otherwise.
#include "../../include/ldpc_mi355x.h"

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

extern "C" void osd_oracle_postprocess(const uint8_t *H, int64_t m, int64_t n, const uint8_t *syndrome,
                                       const uint8_t *bp_err, const double *log_probabs, int64_t osd_order,
                                       uint8_t *out);

namespace ldpc_detail {   // normally provided by ldpc_mi355x.hip
static std::string g_last;
ldpc_status set_error(ldpc_status st, const std::string &msg) { g_last = msg; return st; }
}  // namespace ldpc_detail

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd()
{
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static double uni() { return (double)(rnd() >> 11) / 9007199254740992.0; }

int main()
{
    int cases = 0;
    for (int trial = 0; trial < 60; ++trial) {
        const int64_t m = 1 + (int64_t)(rnd() % 40), n = 1 + (int64_t)(rnd() % 90);
        const double dens = 0.03 + 0.3 * uni();
        std::vector<uint8_t> H((size_t)(m * n));
        for (auto &v : H) v = uni() < dens;
        if (trial % 3 == 0 && m > 2)   // dependent rows: rank deficiency
            for (int64_t j = 0; j < n; ++j) H[(size_t)((m - 1) * n + j)] = H[(size_t)j] ^ H[(size_t)(n + j)];
        if (trial % 5 == 0)            // an empty column and an empty row
            for (int64_t i = 0; i < m; ++i) H[(size_t)(i * n)] = 0;
        std::vector<int64_t> colptr((size_t)n + 1, 0), rowval;
        for (int64_t j = 0; j < n; ++j) {
            for (int64_t i = 0; i < m; ++i)
                if (H[(size_t)(i * n + j)]) rowval.push_back(i);
            colptr[(size_t)j + 1] = (int64_t)rowval.size();
        }
        const int64_t B = 1 + (int64_t)(rnd() % 37);
        std::vector<uint8_t> syn((size_t)(B * m)), bp((size_t)(B * n)), out((size_t)(B * n)), ref((size_t)n);
        std::vector<double> llr((size_t)(B * n));
        for (int64_t b = 0; b < B; ++b) {
            // syndromes in the column space (as BP+OSD sees them) for most rows, arbitrary for some
            std::vector<uint8_t> e((size_t)n);
            for (auto &v : e) v = uni() < 0.1;
            for (int64_t i = 0; i < m; ++i) {
                unsigned p = 0;
                for (int64_t j = 0; j < n; ++j) p ^= (unsigned)(H[(size_t)(i * n + j)] & e[(size_t)j]);
                syn[(size_t)(b * m + i)] = (uint8_t)((b % 7 == 6) ? (rnd() & 1) : p);
            }
            for (int64_t j = 0; j < n; ++j) {
                bp[(size_t)(b * n + j)] = (b % 4 == 3) ? e[(size_t)j] : (uint8_t)(uni() < 0.08);
                // few distinct values => ties in the reliability key; +-Inf and 0 included
                const double vals[] = {-0.1, -2.5, -30.0, 0.0, -INFINITY, -0.6931471805599453, -1e-9};
                llr[(size_t)(b * n + j)] = (trial % 2) ? vals[rnd() % 7] : -8.0 * uni();
            }
        }
        for (int64_t order : {0, 2, 3, 5}) {
            for (int32_t threads : {1, 3}) {
                ldpc_osd *h = nullptr;
                if (ldpc_osd_create(m, n, (int64_t)rowval.size(), colptr.data(), rowval.data(), order, &h) != LDPC_OK) {
                    std::printf("create failed: %s\n", ldpc_detail::g_last.c_str());
                    return 2;
                }
                if (ldpc_osd_postprocess_batch(h, B, syn.data(), bp.data(), llr.data(), out.data(), threads) != LDPC_OK) {
                    std::printf("postprocess failed: %s\n", ldpc_detail::g_last.c_str());
                    return 2;
                }
                ldpc_osd_destroy(h);
                for (int64_t b = 0; b < B; ++b) {
                    osd_oracle_postprocess(H.data(), m, n, &syn[(size_t)(b * m)], &bp[(size_t)(b * n)], &llr[(size_t)(b * n)],
                                           order, ref.data());
                    for (int64_t j = 0; j < n; ++j)
                        if (ref[(size_t)j] != out[(size_t)(b * n + j)]) {
                            std::printf("MISMATCH trial %d m %lld n %lld order %lld threads %d syndrome %lld bit %lld\n", trial,
                                        (long long)m, (long long)n, (long long)order, threads, (long long)b, (long long)j);
                            return 1;
                        }
                    ++cases;
                }
            }
        }
    }
    // argument validation paths
    ldpc_osd *h = nullptr;
    const int64_t cp_bad[] = {0, 2, 1};
    const int64_t rv[] = {0, 1};
    if (ldpc_osd_create(2, 2, 2, cp_bad, rv, 0, &h) == LDPC_OK) { std::printf("bad colptr accepted\n"); return 3; }
    if (ldpc_osd_postprocess_batch(nullptr, 1, nullptr, nullptr, nullptr, nullptr, 1) == LDPC_OK) { std::printf("NULL handle accepted\n"); return 3; }
    std::printf("OK %d syndrome x order x thread cases identical\n", cases);
    return 0;
}
