// stream_stress.cpp - the pipelined schedule stream (ml2048_sched_stream_*, csrc/hostrng.cpp) on an oversubscribed
// host: as many producer workers as the box has processors, plus as many busy "burner" threads again, so that
// workers are preempted at arbitrary points (between publishing a range and anything that follows it - the window
// in which a per-slot completion count used to land on the slot's next occupant and strand the consumer).
// Every delivered schedule is compared with the synchronous ml2048_hostrng_schedule on the same generator state.
// With a fourth argument "slots" the consumer is the zero-copy one (create_pinned + take_slot): it holds up to three
// schedules in caller-provided slot memory, hashes each again when it releases it (a producer that wrote into a held
// slot would change it - and race, which the -fsanitize=thread build reports).
// Built by tests/test_stream_stress.py twice: plain -O2, and -fsanitize=thread without OpenMP (libgomp is not
// instrumented; the stream itself uses std::thread only).  A hang is caught by the test's timeout.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

#include "../../include/ml2048_hip.h"

static uint64_t fnv(const void *p, size_t n, uint64_t h) {
    const uint8_t *b = (const uint8_t *)p;
    for (size_t i = 0; i < n; ++i) h = (h ^ b[i]) * 0x100000001B3ull;
    return h;
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 1500;
    const int n_steps = argc > 2 ? atoi(argv[2]) : 8;
    const int burners = argc > 3 ? atoi(argv[3]) : (int)std::thread::hardware_concurrency();
    ml2048_pcg64 st0 = {0x853C49E6748FEA9Bull, 0xDA3E39CB94B95BDBull, 0x0000000000000001ull, 0x4B47FD7F2E6C1A35ull | 1ull, 0, 0};
    std::vector<uint8_t> perm0(ML2048_RAND_SIZE * 16);
    std::vector<float> flt0(ML2048_RAND_SIZE);
    for (int r = 0; r < ML2048_RAND_SIZE; ++r) {
        for (int c = 0; c < 16; ++c) perm0[r * 16 + c] = (uint8_t)((c + r) & 15);
        flt0[r] = (float)((r * 37) % 101) / 101.0f;
    }
    const int cap = n_steps + 1;
    // reference: the synchronous call, schedule after schedule
    std::vector<uint64_t> want((size_t)iters);
    {
        ml2048_pcg64 st = st0;
        std::vector<uint8_t> perm = perm0, epochs((size_t)cap * ML2048_RAND_SIZE * 16);
        std::vector<float> flt = flt0;
        std::vector<int32_t> sched((size_t)n_steps * 3);
        std::vector<uint32_t> masks((size_t)cap);
        int32_t rand_step = 0;
        for (int k = 0; k < iters; ++k) {
            int32_t E = 0;
            if (ml2048_hostrng_schedule(&st, perm.data(), flt.data(), &rand_step, n_steps, 0.8, sched.data(), epochs.data(),
                                        masks.data(), cap, &E) != 0)
                return puts("sync schedule failed"), 1;
            uint64_t h = fnv(sched.data(), sched.size() * 4, 0xCBF29CE484222325ull);
            h = fnv(epochs.data(), (size_t)E * ML2048_RAND_SIZE * 16, h);
            h = fnv(masks.data(), (size_t)E * 4, h);
            h = fnv(&st, sizeof st, h);
            h = fnv(&rand_step, 4, h);
            want[(size_t)k] = h;
        }
    }
    std::atomic<bool> stop{false};
    std::vector<std::thread> burn;
    for (int b = 0; b < burners; ++b)
        burn.emplace_back([&stop, b] {
            volatile uint64_t x = (uint64_t)b;
            while (!stop.load(std::memory_order_relaxed)) {
                for (int i = 0; i < 20000; ++i) x = x * 6364136223846793005ull + 1442695040888963407ull;
                if ((x >> 60) == 0) std::this_thread::yield();
            }
        });
    const bool slots_mode = argc > 4 && !strcmp(argv[4], "slots");
    int bad = 0;
    uint64_t lcg = 12345;
    if (slots_mode) {
        const int depth = 5;
        const int64_t sb = ml2048_sched_stream_slot_bytes(n_steps);
        std::vector<std::vector<uint8_t>> mem((size_t)depth, std::vector<uint8_t>((size_t)sb + 16));
        void *ptrs[depth];
        for (int d = 0; d < depth; ++d) ptrs[d] = mem[(size_t)d].data() + (16 - (uintptr_t)mem[(size_t)d].data() % 16) % 16;
        ml2048_sched_stream *h = ml2048_sched_stream_create_pinned(&st0, perm0.data(), flt0.data(), 0, n_steps, 0.8, depth, ptrs, sb);
        if (!h) return puts("create_pinned failed"), 1;
        struct Held { int slot; int64_t nbytes; uint64_t hash; };
        std::vector<Held> held;
        auto release_oldest = [&] {
            const Held hd = held.front();
            held.erase(held.begin());
            if (fnv(ptrs[hd.slot], (size_t)hd.nbytes, 1) != hd.hash) puts("a held slot was rewritten"), bad = 1;
            if (ml2048_sched_stream_release(h) != 0) puts("release failed"), bad = 1;
        };
        for (int k = 0; k < iters && !bad; ++k) {
            int32_t slot = -1, E = 0, rs = 0;
            int64_t o_mask = 0, o_tab = 0, nbytes = 0;
            ml2048_pcg64 after;
            if (held.size() >= 3) release_oldest();
            if (ml2048_sched_stream_take_slot(h, &slot, &E, &o_mask, &o_tab, &nbytes, &after, &rs, nullptr, nullptr) != 0) {
                puts("take_slot failed");
                bad = 1;
                break;
            }
            const uint8_t *b = (const uint8_t *)ptrs[slot];
            uint64_t g = fnv(b, (size_t)n_steps * 12, 0xCBF29CE484222325ull);
            g = fnv(b + o_tab, (size_t)E * ML2048_RAND_SIZE * 16, g);
            g = fnv(b + o_mask, (size_t)E * 4, g);
            g = fnv(&after, sizeof after, g);
            g = fnv(&rs, 4, g);
            if (g != want[(size_t)k]) printf("schedule %d differs from the synchronous one\n", k), bad = 1;
            held.push_back({slot, nbytes, fnv(b, (size_t)nbytes, 1)});
            lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
            if ((lcg >> 61) == 0) std::this_thread::sleep_for(std::chrono::microseconds((lcg >> 40) & 255));
            if ((lcg >> 59) == 1 && !held.empty()) release_oldest(); // sometimes run with fewer held
        }
        while (!held.empty() && !bad) release_oldest();
        ml2048_sched_stream_destroy(h);
        stop.store(true);
        for (auto &t : burn) t.join();
        if (bad) return 1;
        printf("stream stress ok (slots): %d schedules of %d steps, %d burner threads\n", iters, n_steps, burners);
        return 0;
    }
    ml2048_sched_stream *h = ml2048_sched_stream_create(&st0, perm0.data(), flt0.data(), 0, n_steps, 0.8, 3);
    if (!h) return puts("create failed"), 1;
    for (int k = 0; k < iters && !bad; ++k) {
        const int32_t *sched;
        const uint8_t *epochs, *perm_after;
        const uint32_t *masks;
        const float *flt_after;
        int32_t E = 0, rs = 0;
        ml2048_pcg64 after;
        if (ml2048_sched_stream_next(h, &sched, &epochs, &masks, &E, &after, &rs, &perm_after, &flt_after) != 0) {
            puts("next failed");
            bad = 1;
            break;
        }
        uint64_t g = fnv(sched, (size_t)n_steps * 12, 0xCBF29CE484222325ull);
        g = fnv(epochs, (size_t)E * ML2048_RAND_SIZE * 16, g);
        g = fnv(masks, (size_t)E * 4, g);
        g = fnv(&after, sizeof after, g);
        g = fnv(&rs, 4, g);
        if (g != want[(size_t)k]) {
            printf("schedule %d differs from the synchronous one\n", k);
            bad = 1;
        }
        if (ml2048_sched_stream_release(h) != 0) {
            puts("release failed");
            bad = 1;
        }
        lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
        if ((lcg >> 61) == 0) std::this_thread::sleep_for(std::chrono::microseconds((lcg >> 40) & 255)); // consumer jitter
    }
    ml2048_sched_stream_destroy(h);
    stop.store(true);
    for (auto &t : burn) t.join();
    if (bad) return 1;
    printf("stream stress ok: %d schedules of %d steps, %d burner threads\n", iters, n_steps, burners);
    return 0;
}
