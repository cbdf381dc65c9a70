// host-side sanitizer run of the native generators (ASan + UBSan); GPU code is not involved
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "tetris_piclim.h"
namespace tpl { int fail_msg(int code, const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); return code; } }
int main() {
    for (int L : {1, 3, 5, 10, 15, 16}) {
        for (int M : {7, 20, 40, 120, 254}) {
            if (M < 2 * L) continue;
            const int64_t n = 300;
            std::vector<uint16_t> rows(n * 20); std::vector<uint8_t> pieces(n * (M + 1)), sol(n * M * 2); std::vector<int32_t> len(n);
            int rc = tpl_generate_configs(L, M, 7, 0, n, 4, 0, rows.data(), pieces.data(), sol.data(), len.data());
            if (rc) { printf("generate L=%d M=%d rc=%d\n", L, M, rc); return 1; }
            std::vector<uint64_t> seeds(40); for (int k = 0; k < 40; ++k) seeds[k] = 1000 + k;
            rc = tpl_generate_configs_pyseed(L, M, seeds.data(), 40, 3, 0, rows.data(), pieces.data(), sol.data(), len.data());
            if (rc) { printf("pyseed L=%d M=%d rc=%d\n", L, M, rc); return 1; }
            // replay the carve of every solution step backwards on a full stack (exercises tpl_carve)
            for (int k = 0; k < 40; ++k) {
                uint16_t b[20]; for (int r = 0; r < 20; ++r) b[r] = r >= 20 - L ? 0x3FF : 0;
                for (int i = len[k] - 1; i >= 0; --i) {
                    int32_t ok = 0;
                    rc = tpl_carve(b, pieces[k * (M + 1) + i], sol[(k * M + i) * 2], sol[(k * M + i) * 2 + 1], i == len[k] - 1, &ok);
                    if (rc || !ok) { printf("carve failed L=%d M=%d k=%d i=%d\n", L, M, k, i); return 1; }
                }
                if (memcmp(b, &rows[k * 20], 40)) { printf("carve replay differs L=%d M=%d k=%d\n", L, M, k); return 1; }
            }
        }
    }
    for (int M : {5, 20, 40, 254}) {
        const int64_t n = 400;
        std::vector<uint64_t> seeds(n); for (int64_t k = 0; k < n; ++k) seeds[k] = k;
        std::vector<uint16_t> rows(n * 20); std::vector<uint8_t> seq(n * M), win(n), sol(n * M * 2), stack(n * M * 3); std::vector<int32_t> failed(n), len(n);
        int rc = tpl_forward_generate(5, M, 4, 1000, seeds.data(), n, 4, rows.data(), seq.data(), win.data(), failed.data(), sol.data(), stack.data(), len.data());
        if (rc) { printf("forward M=%d rc=%d\n", M, rc); return 1; }
    }
    {   // the restart rule's doubled cut-offs and the host pilot's threaded branch: a tight move budget carves, one that does not
        // finish is refused after the pilot (attempts 12-23 on parallel threads), one below the fewest pieces outright
        const int64_t n = 8;
        std::vector<uint16_t> rows(n * 20); std::vector<uint8_t> pieces(n * 41); std::vector<int32_t> len(n);
        int rc = tpl_generate_configs(10, 8, 3, 0, n, 4, 0, rows.data(), pieces.data(), nullptr, len.data());
        if (rc) { printf("tight budget (10, 8) rc=%d\n", rc); return 1; }
        for (int64_t k = 0; k < n; ++k) if (len[k] < 5 || len[k] > 8) { printf("tight budget: solution of %d pieces\n", len[k]); return 1; }
        if (tpl_generate_configs(10, 7, 3, 0, n, 4, 0, rows.data(), pieces.data(), nullptr, nullptr) != TPL_ERR_STATE) { puts("(10, 7) was not refused by the pilot"); return 1; }
        if (tpl_generate_configs(10, 4, 3, 0, n, 4, 0, rows.data(), pieces.data(), nullptr, nullptr) != TPL_ERR_ARG) { puts("(10, 4) was not refused outright"); return 1; }
        if (tpl_generate_configs(10, 40, 3, 0, n, 4, 1, rows.data(), pieces.data(), nullptr, nullptr) != TPL_ERR_STATE) { puts("cutoff 1 was not refused"); return 1; }
    }
    puts("sanitizer run ok");
    return 0;
}
