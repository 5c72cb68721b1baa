// host-side code of the library under ASan + UBSan (no GPU): weight tables, moment fit, export format
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "sg_k1d_host.hpp"
extern "C" void sg_set_error(const char *, ...) {}
extern "C" long savgol_export_header(const SavgolFilter *, const char *, const char *, char *, size_t);
int main()
{
    int fits = 0, exports = 0;
    for (int n = 1; n <= 32; ++n)
        for (int m = 0; m <= 10 && m < 2 * n + 1; ++m)
            for (int d = 0; d <= 4 && d <= m; ++d) {
                if (!sg_weights_valid(n, m, d, 0.5f)) continue;
                SavgolFilter f; memset(&f, 0, sizeof f);
                f.config.half_window = (uint8_t)n; f.config.poly_order = (uint8_t)m; f.config.derivative = (uint8_t)d; f.config.time_step = 0.5f;
                sg_weights_fill(&f);
                float table[sg::MOMENT_TABLE_FLOATS];
                fits += sg1d_momenth_prepare(n, f.center_weights, table) != 0;
                std::vector<char> buf(1);
                const long need = savgol_export_header(&f, (n & 1) ? "pfx" : nullptr, "T", buf.data(), 1);
                buf.resize((size_t)need + 1);
                if (savgol_export_header(&f, (n & 1) ? "pfx" : nullptr, "T", buf.data(), buf.size()) != need || strlen(buf.data()) != (size_t)need) { printf("export mismatch\n"); return 1; }
                ++exports;
            }
    for (int nx = 1; nx <= 16; nx += 3) for (int ny = 1; ny <= 16; ny += 5) for (int o = 0; o <= 6; ++o) {
        Savgol2DConfig c; memset(&c, 0, sizeof c);
        c.half_window_x = (uint8_t)nx; c.half_window_y = (uint8_t)ny; c.poly_order = (uint8_t)o; c.deriv_x = (uint8_t)(o > 1); c.delta_x = 0.5f; c.delta_y = 2.0f;
        if (!sg2d_config_ok(&c)) continue;
        std::vector<float> W((size_t)(2 * nx + 1) * (2 * ny + 1)); double coef[64];
        sg2d_weights_fill(&c, W.data(), coef);
    }
    printf("sanitizer pass: %d moment fits, %d exports\n", fits, exports);
    return 0;
}
