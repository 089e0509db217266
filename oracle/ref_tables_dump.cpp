// oracle/_ref builder input: dumps the signal tables straight from the reference header
// (/root/reference/include/constants.h, the only reference file that compiles in this image without
// stand-ins: it includes just <cstddef>/<cstdint>).  Used by tests/test_tables.py to check
// galileo-sdr-sim_amd/csrc/e1_tables.inc and the oracle's expansions.  Built only where
// /root/reference exists; the binary lands in oracle/_ref/ (git-ignored).
#include <cstdio>
#include <cstring>
#include "constants.h"

int main()
{
    std::printf("cos");
    for (int k = 0; k < 512; k++) std::printf(" %d", cosTable512[k]);
    std::printf("\nsin");
    for (int k = 0; k < 512; k++) std::printf(" %d", sinTable512[k]);
    std::printf("\ncs25");
    for (int k = 0; k < 25; k++) std::printf(" %d", (int)GALILEO_E1_SECONDARY_CODE[k]);
    std::printf("\n");
    for (int p = 0; p < GALILEO_E1_NUMBER_OF_CODES; p++) {
        std::printf("e1b %d %zu %s\n", p + 1, std::strlen(GALILEO_E1_B_PRIMARY_CODE[p]), GALILEO_E1_B_PRIMARY_CODE[p]);
        std::printf("e1c %d %zu %s\n", p + 1, std::strlen(GALILEO_E1_C_PRIMARY_CODE[p]), GALILEO_E1_C_PRIMARY_CODE[p]);
    }
    std::printf("consts %d %d %d %.17g %.17g %.17g\n", MAX_CHAN, CA_SEQ_LEN_E1, N_SYM_PAGE, (double)SAMP_RATE,
                (double)CODE_FREQ_E1, (double)CARR_TO_CODE_E1);
    return 0;
}
