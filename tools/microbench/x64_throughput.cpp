// Per-instruction throughput of the scalar Keccak's instruction mix on this core (independent chains, registers only):
// how many xor / rorx / andn per cycle the integer cluster sustains, alone and in Keccak's 76 : 29 : 25 proportion.
#include <chrono>
#include <cstdint>
#include <cstdio>
static volatile uint64_t g_sink;
#define RUN(NAME, BODY, NINSTR) { \
    uint64_t a = 1, b = 2, c = 3, d = 4, e = 5, f = 6, g = 7, h = 8, i_ = 9, j = 10, k = 11, l = 12; \
    const int iters = 20000000; \
    const auto t0 = std::chrono::steady_clock::now(); \
    for (int it = 0; it < iters; ++it) asm volatile(BODY : "+r"(a), "+r"(b), "+r"(c), "+r"(d), "+r"(e), "+r"(f), "+r"(g), "+r"(h), "+r"(i_), "+r"(j), "+r"(k), "+r"(l)); \
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); \
    g_sink = a ^ b ^ c ^ d ^ e ^ f ^ g ^ h ^ i_ ^ j ^ k ^ l; \
    printf("%-44s %.3f ns per instruction  (%.2f per ns)\n", NAME, dt / iters / (NINSTR) * 1e9, (NINSTR) * (double)iters / dt / 1e9); }
int main() {
    RUN("xor r,r  x12 independent", "xor %1,%0\n xor %2,%1\n xor %3,%2\n xor %4,%3\n xor %5,%4\n xor %6,%5\n xor %7,%6\n xor %8,%7\n xor %9,%8\n xor %10,%9\n xor %11,%10\n xor %0,%11", 12)
    RUN("xor r,r  x12 (6 chains of 2)", "xor %6,%0\n xor %7,%1\n xor %8,%2\n xor %9,%3\n xor %10,%4\n xor %11,%5\n xor %6,%0\n xor %7,%1\n xor %8,%2\n xor %9,%3\n xor %10,%4\n xor %11,%5", 12)
    RUN("rorx x12 independent", "rorx $7,%0,%0\n rorx $7,%1,%1\n rorx $7,%2,%2\n rorx $7,%3,%3\n rorx $7,%4,%4\n rorx $7,%5,%5\n rorx $7,%6,%6\n rorx $7,%7,%7\n rorx $7,%8,%8\n rorx $7,%9,%9\n rorx $7,%10,%10\n rorx $7,%11,%11", 12)
    RUN("andn x12 independent", "andn %6,%0,%0\n andn %7,%1,%1\n andn %8,%2,%2\n andn %9,%3,%3\n andn %10,%4,%4\n andn %11,%5,%5\n andn %0,%6,%6\n andn %1,%7,%7\n andn %2,%8,%8\n andn %3,%9,%9\n andn %4,%10,%10\n andn %5,%11,%11", 12)
    RUN("mix 7 xor : 3 rorx : 2 andn", "xor %6,%0\n rorx $7,%1,%1\n xor %7,%2\n andn %8,%3,%3\n xor %9,%4\n rorx $9,%5,%5\n xor %0,%6\n xor %2,%7\n andn %4,%8,%8\n xor %10,%9\n rorx $3,%10,%10\n xor %1,%11", 12)
    return 0;
}
