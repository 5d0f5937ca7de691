#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "hk.h"
#include <thread>
#include <atomic>
#include <chrono>
int main(int argc, char **argv) {
    const int threads = argc > 1 ? atoi(argv[1]) : 8;
    // synthetic tags: 6 balls on parabolas + clutter tracks + noise, 400 frames
    const int F = 400, W = 1280, H = 720;
    std::vector<int> frame; std::vector<double> x, y, r, tw;
    unsigned s = 12345;
    auto rnd = [&] { s = s * 1664525u + 1013904223u; return (s >> 8) / 16777216.0; };
    for (int f = 0; f < F; ++f) {
        for (int b = 0; b < 6; ++b) {
            const double period = 40 + 3 * b, t = fmod(f + 7.0 * b, period) - period / 2;
            const double g = 0.9;
            frame.push_back(f); x.push_back(200 + 140 * b + 2.0 * t + rnd());
            y.push_back(150 + 0.5 * g * t * t + rnd()); r.push_back(16 + 0.3 * rnd()); tw.push_back(1.0);
        }
        for (int d = 0; d < 12; ++d) {
            frame.push_back(f); x.push_back(60 + 100 * d + 3 * sin(f * 0.05 + d) + rnd());
            y.push_back(100 + 45 * d + 3 * cos(f * 0.04 + d) + rnd()); r.push_back(15 + rnd()); tw.push_back(1.0);
        }
        for (int n = 0; n < 4; ++n) {
            frame.push_back(f); x.push_back(W * rnd()); y.push_back(H * rnd()); r.push_back(14 + 4 * rnd()); tw.push_back(1.0);
        }
    }
    hk_scan_params p;
    hk_scan_params_default(&p, H, H, 60.0);
    hk_set_host_threads(threads);
    std::atomic<bool> stop{false};
    std::vector<std::thread> helpers;
    const int nh = argc > 2 ? atoi(argv[2]) : 0;
    for (int h = 0; h < nh; ++h)
        helpers.emplace_back([&] {
            while (!stop.load())
                if (!hk_step3_help(256)) std::this_thread::sleep_for(std::chrono::microseconds(200));
        });
    for (int rep = 0; rep < 3; ++rep) {
        hk_clip *c = nullptr;
        if (hk_clip_create(W, H, 60.0, F, &p, (int)frame.size(), frame.data(), x.data(), y.data(), r.data(), tw.data(), 0, &c) != 0) {
            printf("create failed: %s\n", hk_last_error()); return 1;
        }
        if (hk_clip_build_initial_arcs(c) != 0) { printf("build failed: %s\n", hk_last_error()); return 1; }
        printf("threads %d: %d arcs\n", threads, hk_clip_arc_count(c));
        hk_clip_destroy(c);
    }
    stop.store(true);
    for (auto &t : helpers) t.join();
    return 0;
}
