// test_stream.cpp -- CPU-only check of the C++ twins of cpi_amd/stream.py (parse_imu_text, assemble_windows):
// reads an IMU text file and update times, prints the assembled knots/first/count for the Python test to compare.
#include <cstdio>
#include <fstream>
#include <sstream>

#include "../../cpi_amd/csrc/cpi_host.hpp"

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    std::ifstream f(argv[1]);
    std::stringstream ss; ss << f.rdbuf();
    std::vector<double> stream = cpi_host::parse_imu_text(ss.str());
    std::vector<double> ut;
    std::ifstream g(argv[2]);
    double t;
    while (g >> t) ut.push_back(t);
    cpi_host::WindowSet ws = cpi_host::assemble_windows(stream, ut);
    printf("%zu %zu %d\n", stream.size() / 7, ws.first.size(), ws.max_count);
    for (size_t i = 0; i < ws.first.size(); i++) printf("%lld %d\n", (long long)ws.first[i], ws.count[i]);
    for (size_t i = 0; i < ws.knots.size(); i++) printf("%.17g%c", ws.knots[i], (i % 7 == 6) ? '\n' : ' ');
    return 0;
}
