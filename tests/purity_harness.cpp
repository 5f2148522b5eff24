// Test driver for cli/cli_purity.h (tests/test_purity_cpu.py): "n initial f0..f4" then n lines "ratio count" on stdin -> the purity on stdout, the
// report in <argv[1]>_purity.out, the reference's error messages on stderr.
#include "../longphase-s_amd/cli/cli_purity.h"
int main(int argc, char **argv) {
    if (argc < 2) return 2;
    std::vector<PurityDatum> v; size_t n, initial; int first_filters[5];
    if (scanf("%zu %zu %d %d %d %d %d", &n, &initial, &first_filters[0], &first_filters[1], &first_filters[2], &first_filters[3], &first_filters[4]) != 7) return 2;
    for (size_t i = 0; i < n; ++i) { double r; int c; if (scanf("%lf %d", &r, &c) != 2) return 2; v.push_back({r, c}); }
    printf("%.17g\n", estimate_purity(v, initial, first_filters, argv[1]));
    return 0;
}
