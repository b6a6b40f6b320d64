// Host check of ContactFused<TwoBody> (bepu_device_constraints.h) against Contact<N, TwoBody> for N = 1..4: same bits out of warm start, solve and the incremental
// depth update on random inputs, with negative depths, zero weights (the fallback branch of the friction centre) and zero impulses mixed in.
// Built and run by tests/test_contact_fused_host.py (clang++ -ffp-contract=off, x86 host; no GPU).
#define BEPU_PIN_ENABLED 0
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include "bepu_device_constraints.h"

using namespace bd;

struct HostGate {
    static constexpr bool kPin = false;
    void operator()(BodyVel&, BodyVel&) const {}
    void many(BodyVel*) const {}
};

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t next_u32() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (uint32_t)(rng_state >> 32); }
static float uniform(float lo, float hi) { return lo + (hi - lo) * (float)(next_u32() >> 8) * (1.0f / 16777216.0f); }

static Inertia random_inertia() {
    Inertia i;
    const float a = uniform(0.2f, 3.0f), b = uniform(0.2f, 3.0f), c = uniform(0.2f, 3.0f);
    i.t = {a, uniform(-0.1f, 0.1f), b, uniform(-0.1f, 0.1f), uniform(-0.1f, 0.1f), c};
    i.invMass = uniform(0.1f, 2.0f);
    return i;
}
static BodyVel random_velocity() { return {{uniform(-2, 2), uniform(-2, 2), uniform(-2, 2)}, {uniform(-3, 3), uniform(-3, 3), uniform(-3, 3)}}; }

template <int N, bool TwoBody>
static long check(int rounds, int mode) {
    using T = Contact<N, TwoBody>;
    using F = ContactFused<TwoBody>;
    long bad = 0;
    for (int r = 0; r < rounds; ++r) {
        float p[T::prestepFloats], a[T::impulseFloats];
        for (int i = 0; i < N; ++i) {
            p[4 * i] = uniform(-1, 1); p[4 * i + 1] = uniform(-1, 1); p[4 * i + 2] = uniform(-1, 1);
            const int kind = mode == 2 ? 1 : (int)(next_u32() % 4);
            p[4 * i + 3] = kind == 0 ? uniform(0.0f, 0.05f) : (kind == 1 ? uniform(-0.05f, -0.001f) : (kind == 2 ? 0.0f : uniform(-0.02f, 0.02f)));
        }
        int at = 4 * N;
        if (TwoBody) { p[at++] = uniform(-1, 1); p[at++] = uniform(-1, 1); p[at++] = uniform(-1, 1); }
        float nx = uniform(-1, 1), ny = uniform(-1, 1), nz = uniform(-1, 1);
        const float len = std::sqrt(nx * nx + ny * ny + nz * nz) + 1e-6f;
        p[at++] = nx / len; p[at++] = ny / len; p[at++] = nz / len;
        p[at++] = uniform(0.1f, 1.5f); p[at++] = uniform(10.0f, 200.0f); p[at++] = uniform(0.5f, 3.0f); p[at++] = uniform(0.5f, 4.0f);
        for (int i = 0; i < T::impulseFloats; ++i) a[i] = (mode == 1 && (next_u32() & 1)) ? 0.0f : uniform(i >= 2 && i < 2 + N ? 0.0f : -1.0f, 1.0f);
        const Inertia iA = random_inertia(), iB = random_inertia();
        const BodyVel vA0 = random_velocity(), vB0 = random_velocity();
        const float dt = 1.0f / 240.0f, inverseDt = 240.0f;
        // Contact4 layout of the same lane
        float fp[F::prestepFloats], fa[F::impulseFloats];
        // rows of contacts the lane does not have: whatever the loader left there (zeros in the kernel; here zeros, garbage or NaN) must not reach a result
        const int absent = (int)(next_u32() % 3);
        for (int i = 0; i < F::prestepFloats; ++i) fp[i] = absent == 0 ? 0.0f : (absent == 1 ? uniform(-5, 5) : std::nanf(""));
        for (int i = 0; i < F::impulseFloats; ++i) fa[i] = absent == 0 ? 0.0f : (absent == 1 ? uniform(-5, 5) : std::nanf(""));
        for (int i = 0; i < 4 * N; ++i) fp[i] = p[i];
        for (int i = 0; i < F::commonFloats; ++i) fp[16 + i] = p[4 * N + i];
        fa[0] = a[0]; fa[1] = a[1];
        for (int i = 0; i < N; ++i) fa[2 + i] = a[2 + i];
        fa[6] = a[2 + N];
        for (int stage = 0; stage < 3; ++stage) {
            float p1[T::prestepFloats], a1[T::impulseFloats], p2[F::prestepFloats], a2[F::impulseFloats];
            memcpy(p1, p, sizeof(p)); memcpy(a1, a, sizeof(a)); memcpy(p2, fp, sizeof(fp)); memcpy(a2, fa, sizeof(fa));
            BodyVel vA1 = vA0, vB1 = vB0, vA2 = vA0, vB2 = vB0;
            HostGate gate;
            if (stage == 0) {
                T::warmStart(V3{}, Q{}, iA, V3{}, Q{}, iB, p1, a1, vA1, vB1, gate);
                F::warmStart(iA, iB, p2, a2, N, vA2, vB2, gate);
            } else if (stage == 1) {
                T::solve(V3{}, Q{}, iA, V3{}, Q{}, iB, dt, inverseDt, p1, a1, vA1, vB1, gate);
                F::solve(iA, iB, dt, inverseDt, p2, a2, N, vA2, vB2, gate);
            } else {
                T::incrementalUpdate(dt, vA1, vB1, p1);
                F::incrementalUpdate(dt, vA2, vB2, p2, N);
            }
            bool same = memcmp(&vA1, &vA2, sizeof(BodyVel)) == 0 && (!TwoBody || memcmp(&vB1, &vB2, sizeof(BodyVel)) == 0);
            same = same && memcmp(a1, a2, 2 * sizeof(float)) == 0 && memcmp(a1 + 2, a2 + 2, N * sizeof(float)) == 0 && memcmp(a1 + 2 + N, a2 + 6, sizeof(float)) == 0;
            for (int i = 0; i < N; ++i) same = same && memcmp(&p1[4 * i + 3], &p2[4 * i + 3], sizeof(float)) == 0;
            if (!same) { if (bad < 5) fprintf(stderr, "mismatch: N=%d two_body=%d stage=%d round=%d mode=%d\n", N, (int)TwoBody, stage, r, mode); ++bad; }
        }
    }
    return bad;
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 20000;
    long bad = 0;
    for (int mode = 0; mode < 3; ++mode) {
        bad += check<1, true>(rounds, mode) + check<2, true>(rounds, mode) + check<3, true>(rounds, mode) + check<4, true>(rounds, mode);
        bad += check<1, false>(rounds, mode) + check<2, false>(rounds, mode) + check<3, false>(rounds, mode) + check<4, false>(rounds, mode);
    }
    printf("contact_fused_host: %d rounds x 3 modes x 8 types x 3 stages, %ld mismatches\n", rounds, bad);
    return bad ? 1 : 0;
}
