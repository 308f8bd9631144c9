// advec1d: the reference's CPU-runnable configuration (src/advec1d/main.cpp:35-122) with N and K
// as arguments. Host only; prints "Error: <max-norm error>" like the reference.
//   ./bin/advec1d [N=4] [K=30] [finalTime=20]
#include "blitzdg/Advec1d.hpp"
#include <cstdlib>
#include <iostream>

int main(int argc, char** argv) {
    using namespace blitzdg;
    const index_type N = argc > 1 ? std::atoi(argv[1]) : 4, K = argc > 2 ? std::atoi(argv[2]) : 30;
    const real_type T = argc > 3 ? std::atof(argv[3]) : 20.0;
    try {
        index_type steps = 0;
        const real_type err = advec1d::run(N, K, -1.0, 4.0, 0.1, 0.8, T, &steps);
        std::cout << "steps: " << steps << "\nError: " << err << std::endl;
    } catch (const std::exception& e) {
        std::cerr << "error: " << e.what() << "\n";
        return 1;
    }
    return 0;
}
