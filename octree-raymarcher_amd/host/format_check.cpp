// format_check.cpp — host-side checks of csrc/svo_format.h that need no device (tests/test_host_units.py).
//   g++ -std=c++17 -I../csrc format_check.cpp -o format_check
// The device fill's per-node word (fill_pack: 3-bit action + child-block index): every index the fill admits - up to
// FILL_KIDS_LIMIT - 1 = 2^28 - 1, which builder.hip's frontier guard enforces - must come back whole, with every action.  Until
// round 4 the index shared the word with an 8-bit action field and lost its top bits from 2^24 child blocks on (a depth-15
// chunk with an exposed water plane gets there).
#include <cstdint>
#include <cstdio>

#include "svo_format.h"

int main()
{
    using namespace svo;
    const uint64_t probes[] = { 0, 1, 255, (1ull << 16) + 3, (1ull << 24) - 1, 1ull << 24, (1ull << 24) + 12345, 1ull << 27, FILL_KIDS_LIMIT - 2, FILL_KIDS_LIMIT - 1 };
    int bad = 0;
    for (uint64_t kids : probes)
        for (uint32_t a = 0; a < (1u << FILL_ACTION_BITS); ++a) {
            const uint32_t w = fill_pack(a, (uint32_t)kids);
            if (fill_action(w) != a || fill_kids(w) != kids) { std::printf("fill_pack(%u, %llu) -> action %u, kids %u\n", a, (unsigned long long)kids, fill_action(w), fill_kids(w)); ++bad; }
        }
    // the guard's limit is exactly what the word can hold: 8 list entries per child block stay below 2^31
    if (FILL_KIDS_LIMIT * 8 != (1ull << 31) || (FILL_KIDS_LIMIT - 1) >> (32 - FILL_ACTION_BITS) != 0) { std::printf("FILL_KIDS_LIMIT does not match the packing\n"); ++bad; }
    std::printf("format_check: %d failures\n", bad);
    return bad ? 1 : 0;
}
