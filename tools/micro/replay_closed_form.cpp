// Closed form of the integration kernel's z-prefix replay, as a CPU prototype with its own check.
//
// The reference integrateKernel advances a column's camera-space position by `acc += d` once per z step, in
// binary32 with round-to-nearest-even.  The HIP kernel replays those additions one by one to reach a box in the
// middle of a column.  This file states the alternative that was weighed against the replay: while the accumulator
// stays inside one binade its unit in the last place u is fixed, acc = m * u with an integer m, and every addition
// moves m by the same integer q = round(d / u) (a tie d / u = k + 1/2 resolves the same way from the first even m
// on).  So the state after k steps inside the binade is m + k * q, and only the step that leaves the binade has to
// be a real addition.
//
//   replay_closed_form(a, d, n)   state after n additions, by jumps; *phases counts jumps + real additions
//   replay_sequential(a, d, n)    the n additions
//
// The program compares the two bit for bit on random and adversarial inputs (ties, binade edges, sign changes,
// increments below half a unit, the column geometry of a camera inside an 8 m volume of 512 to 4096 planes) and
// prints how many phases a replay needs, which is what decides whether the closed form is cheaper than the loop.
//
// build: g++ -O2 -ffp-contract=off -o replay_closed_form replay_closed_form.cpp ; run: ./replay_closed_form [cases [wavefronts per volume size]]
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>

static uint32_t bits(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}

static float replay_sequential(float a, float d, int n) {
    for (int i = 0; i < n; ++i) {
        volatile float s = a + d;      // one binary32 rounding per step, whatever the compiler would like to keep
        a = s;
    }
    return a;
}

static float replay_closed_form(float a, float d, int n, int *phases) {
    int ph = 0;
    while (n > 0) {
        bool jumped = false;
        if (d != 0.0f && std::isfinite(a) && std::isfinite(d) && std::fabs(a) >= 1.17549435e-38f) {
            int e;                                         // a = f * 2^e with 0.5 <= |f| < 1
            std::frexp(a, &e);
            const double u = std::ldexp(1.0, e - 24);     // unit in the last place of a
            const double m = (double)a / u;                // signed integer, 2^23 <= |m| < 2^24
            const double dq = (double)d / u;               // exact: a power-of-two scaling
            if (std::fabs(dq) < 9.0e15) {
                const double fl = std::floor(dq);
                const bool tie = dq - fl == 0.5;
                double q;
                bool usable = true;
                if (!tie) {
                    q = std::nearbyint(dq);                // round(m + dq) = m + round(dq) for an integer m
                } else {
                    // the sum lies half-way between two integers and goes to the even one: once m is even the
                    // even one of floor(dq), floor(dq) + 1 is added every time; an odd m takes one real step first
                    usable = std::fmod(std::fabs(m), 2.0) == 0.0;
                    q = std::fmod(std::fabs(fl), 2.0) == 0.0 ? fl : fl + 1.0;
                }
                const double am = std::fabs(m);
                // an accumulator that sits on the power of two itself and is pushed towards zero is rounded on the
                // finer grid below it: a real step
                if (am == 8388608.0 && (dq > 0) != (m > 0)) usable = false;
                if (usable) {
                    if (q == 0.0) {                        // the increment is below half a unit: nothing ever changes
                        if (phases) *phases = ph + 1;
                        return a;
                    }
                    const double aq = std::fabs(q);
                    const bool outward = (q > 0) == (m > 0);
                    // steps after which |m| is still inside the binade, with the sums themselves on this grid:
                    // outward up to 2^24 - 1; inward down to 2^23 + 1 (just below 2^23 the grid is twice as fine)
                    double k = outward ? std::floor((16777215.0 - am) / aq) : std::floor((am - 8388609.0) / aq);
                    if (k >= 1.0) {
                        if (k > (double)n) k = (double)n;
                        a = (float)((m + k * q) * u);      // exact: an integer below 2^24 times a power of two
                        n -= (int)k;
                        jumped = true;
                    }
                }
            }
        }
        if (!jumped) {
            volatile float s = a + d;
            a = s;
            --n;
        }
        ++ph;
    }
    if (phases) *phases = ph;
    return a;
}

struct Tally {
    long cases = 0, mismatches = 0;
    double phases = 0, steps = 0;
    int worst_phases = 0;
};

static void check(float a, float d, int n, Tally &t) {
    int ph = 0;
    const float want = replay_sequential(a, d, n);
    const float got = replay_closed_form(a, d, n, &ph);
    if (bits(want) != bits(got) && !(std::isnan(want) && std::isnan(got))) {
        if (t.mismatches < 10)
            std::printf("MISMATCH a=%a d=%a n=%d sequential=%a closed=%a\n", a, d, n, want, got);
        ++t.mismatches;
    }
    ++t.cases;
    t.phases += ph;
    t.steps += n;
    if (ph > t.worst_phases) t.worst_phases = ph;
}

int main(int argc, char **argv) {
    const long cases = argc > 1 ? std::atol(argv[1]) : 200000;
    std::mt19937_64 rng(12345);
    std::uniform_real_distribution<double> uni(0.0, 1.0);
    Tally any, ties, edges, tiny;
    for (long i = 0; i < cases; ++i) {
        // anything: magnitudes over sixty binades, both signs, up to 4096 steps
        const float a = (float)std::ldexp(uni(rng) * 2 - 1, (int)(uni(rng) * 60) - 30);
        const float d = (float)std::ldexp(uni(rng) * 2 - 1, (int)(uni(rng) * 60) - 30);
        check(a, d, 1 + (int)(uni(rng) * 4096), any);
        // ties: increments with a few mantissa bits against accumulators near a power of two
        const int e = (int)(uni(rng) * 16) - 8;
        const float at = (float)std::ldexp(1.0 + std::floor(uni(rng) * 64) / 8388608.0, e) * (uni(rng) < 0.5 ? -1.f : 1.f);
        const float dt = (float)std::ldexp(std::floor(1 + uni(rng) * 15), e - 24 - (int)(uni(rng) * 3)) * (uni(rng) < 0.5 ? -1.f : 1.f);
        check(at, dt, 1 + (int)(uni(rng) * 4096), ties);
        // binade edges: start a few units from a power of two, walk across it in either direction
        const float ae = (float)std::ldexp(1.0, e) * (1.0f + (float)((int)(uni(rng) * 9) - 4) * 1.1920929e-7f);
        const float de = (float)std::ldexp(uni(rng) * 2 - 1, e - 20 - (int)(uni(rng) * 6));
        check(uni(rng) < 0.5 ? ae : -ae, de, 1 + (int)(uni(rng) * 512), edges);
        // increments around half a unit of the accumulator
        const float as = (float)std::ldexp(1.0 + uni(rng), e);
        const float ds = (float)std::ldexp(0.25 + uni(rng) * 0.5, e - 23) * (uni(rng) < 0.5 ? -1.f : 1.f);
        check(as, ds, 1 + (int)(uni(rng) * 4096), tiny);
    }
    std::printf("anything      : %ld cases, %ld mismatches, %.1f phases for %.0f steps on average, worst %d phases\n",
                any.cases, any.mismatches, any.phases / any.cases, any.steps / any.cases, any.worst_phases);
    std::printf("ties          : %ld cases, %ld mismatches, %.1f phases, worst %d\n", ties.cases, ties.mismatches,
                ties.phases / ties.cases, ties.worst_phases);
    std::printf("binade edges  : %ld cases, %ld mismatches, %.1f phases, worst %d\n", edges.cases, edges.mismatches,
                edges.phases / edges.cases, edges.worst_phases);
    std::printf("half a unit   : %ld cases, %ld mismatches, %.1f phases, worst %d\n", tiny.cases, tiny.mismatches,
                tiny.phases / tiny.cases, tiny.worst_phases);

    // The kernel's own use: a camera somewhere inside an 8 m cube looking anywhere; the six accumulators of a column
    // (camera-space position and its image of the intrinsic matrix) replayed to a random plane.  A wavefront waits
    // for its slowest lane and a lane for its slowest accumulator, so the figure of merit is the largest phase count
    // among the 6 accumulators of 64 neighbouring columns.
    long geo_mismatch = 0;
    for (int planes = 512; planes <= 4096; planes *= 2) {
        double wave_phases = 0, wave_steps = 0;
        int worst = 0;
        const int waves = argc > 2 ? std::atoi(argv[2]) : 2000;
        for (int w = 0; w < waves; ++w) {
            double ax = uni(rng) * 2 - 1, ay = uni(rng) * 2 - 1, az = uni(rng) * 2 - 1, ang = uni(rng) * 3.0;
            const double nrm = std::sqrt(ax * ax + ay * ay + az * az) + 1e-12;
            ax /= nrm; ay /= nrm; az /= nrm;
            const double c = std::cos(ang), s = std::sin(ang), C = 1 - c;
            const float R[9] = {(float)(c + ax * ax * C), (float)(ax * ay * C - az * s), (float)(ax * az * C + ay * s),
                                (float)(ay * ax * C + az * s), (float)(c + ay * ay * C), (float)(ay * az * C - ax * s),
                                (float)(az * ax * C - ay * s), (float)(az * ay * C + ax * s), (float)(c + az * az * C)};
            const float cam[3] = {(float)(1 + uni(rng) * 6), (float)(1 + uni(rng) * 6), (float)(1 + uni(rng) * 6)};
            const float fx = 591.1f, fy = 590.1f, cx = 331.0f, cy = 234.6f;
            const float vz = 8.0f / (float)planes;
            const int n = 1 + (int)(uni(rng) * (planes - 1));
            const int y = (int)(uni(rng) * planes), x0 = 64 * (int)(uni(rng) * (planes / 64));
            int wave_worst = 0;
            for (int lane = 0; lane < 64; ++lane) {
                const float p[3] = {((float)(x0 + lane) + 0.5f) * vz, ((float)y + 0.5f) * vz, 0.5f * vz};
                float pos[3], del[3];
                for (int r = 0; r < 3; ++r) {      // inverse pose = transpose(R) * (p - cam)
                    pos[r] = R[0 + r] * (p[0] - cam[0]) + R[3 + r] * (p[1] - cam[1]) + R[6 + r] * (p[2] - cam[2]);
                    del[r] = R[6 + r] * vz;
                }
                const float acc[6] = {pos[0], pos[1], pos[2], fx * pos[0] + cx * pos[2], fy * pos[1] + cy * pos[2], pos[2]};
                const float inc[6] = {del[0], del[1], del[2], fx * del[0] + cx * del[2], fy * del[1] + cy * del[2], del[2]};
                for (int q = 0; q < 6; ++q) {
                    int ph = 0;
                    const float got = replay_closed_form(acc[q], inc[q], n, &ph);
                    if (bits(got) != bits(replay_sequential(acc[q], inc[q], n))) ++geo_mismatch;
                    if (ph > wave_worst) wave_worst = ph;
                }
            }
            wave_phases += wave_worst;
            wave_steps += n;
            if (wave_worst > worst) worst = wave_worst;
        }
        std::printf("%4d planes    : a wavefront replays %.0f steps on average; its slowest accumulator needs %.1f phases "
                    "on average, %d at worst\n", planes, wave_steps / waves, wave_phases / waves, worst);
    }
    std::printf("column geometry: %ld mismatches\n", geo_mismatch);
    const long bad = any.mismatches + ties.mismatches + edges.mismatches + tiny.mismatches + geo_mismatch;
    std::printf("%s\n", bad == 0 ? "ALL EQUAL" : "DIFFERENCES FOUND");
    return bad == 0 ? 0 : 1;
}
