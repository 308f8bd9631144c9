// burgers1d: the reference's 1-D viscous Burgers driver (src/burgers1d/main.cpp:28-115) with N and K as arguments.
// Host only; prints "Error: <max-norm error against the travelling wave>" like the reference.
//   ./bin/burgers1d [N=6] [K=40] [finalTime=0.1]
#include "blitzdg/Burgers1d.hpp"
#include <cstdlib>
#include <iostream>

int main(int argc, char** argv) {
    using namespace blitzdg;
    const index_type N = argc > 1 ? std::atoi(argv[1]) : 6, K = argc > 2 ? std::atoi(argv[2]) : 40;
    const real_type T = argc > 3 ? std::atof(argv[3]) : 0.1;
    try {
        index_type steps = 0;
        const real_type err = burgers1d::run(N, K, -5.0, 5.0, 1.0, 0.1, 0.5, 0.75, T, &steps);
        std::cout << "steps: " << steps << "\nError: " << err << std::endl;
    } catch (const std::exception& e) {
        std::cerr << "error: " << e.what() << "\n";
        return 1;
    }
    return 0;
}
