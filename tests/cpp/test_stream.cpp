// test_stream.cpp -- CPU-only check of the C++ twins of cpi_amd/stream.py (parse_imu_text, assemble_windows,
// assemble_windows_tiled): reads an IMU text file and update times, prints the assembled knots/first/count -- or, with a
// third argument "tiled", W N, the counts and the tile array -- for the Python test to compare.
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
    if (argc > 3 && std::string(argv[3]) == "tiled") {
        cpi_host::TiledWindowSet ts = cpi_host::assemble_windows_tiled(stream, ut);
        printf("%lld %d\n", (long long)ts.W, ts.N);
        for (size_t i = 0; i < ts.count.size(); i++) printf("%d\n", ts.count[i]);
        for (size_t i = 0; i < ts.tiles.size(); i++) printf("%.17g%c", ts.tiles[i], (i % 64 == 63) ? '\n' : ' ');
        return 0;
    }
    cpi_host::WindowSet ws = cpi_host::assemble_windows(stream, ut);
    printf("%zu %zu %d\n", stream.size() / 7, ws.first.size(), ws.max_count);
    for (size_t i = 0; i < ws.first.size(); i++) printf("%lld %d\n", (long long)ws.first[i], ws.count[i]);
    for (size_t i = 0; i < ws.knots.size(); i++) printf("%.17g%c", ws.knots[i], (i % 7 == 6) ? '\n' : ' ');
    return 0;
}
