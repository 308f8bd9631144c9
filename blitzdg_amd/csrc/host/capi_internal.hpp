// Internal definitions shared by the C-ABI translation units: the concrete
// handle types and the exception -> status-code guard.
#pragma once
#include "blitzdg_hip.h"
#include "blitzdg/MeshManager.hpp"
#include "blitzdg/Nodes1DProvisioner.hpp"
#include "blitzdg/TriangleNodesProvisioner.hpp"
#include <exception>
#include <stdexcept>
#include <string>

struct bdg_mesh {
    blitzdg::MeshManager mgr;
};

struct bdg_trinodes {
    blitzdg::TriangleNodesProvisioner prov;
    bool hasFilter = false;
};

struct bdg_gaussctx {
    blitzdg::GaussFaceContext2D ctx;
};

struct bdg_cubctx {
    blitzdg::CubatureContext2D ctx;
};

struct bdg_nodes1d {
    blitzdg::Nodes1DProvisioner prov;
};

namespace bdg_detail {

void set_error(const std::string& msg);

struct arg_error : std::invalid_argument {
    using std::invalid_argument::invalid_argument;
};
struct hip_error : std::runtime_error {
    using std::runtime_error::runtime_error;
};
struct unstable_error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

/// Runs fn, maps exceptions to BDG_ERR_* and records the message.
template <typename Fn>
int guard(Fn&& fn) noexcept {
    try {
        fn();
        return BDG_OK;
    } catch (const arg_error& e) {
        set_error(e.what());
        return BDG_ERR_ARGUMENT;
    } catch (const hip_error& e) {
        set_error(e.what());
        return BDG_ERR_HIP;
    } catch (const unstable_error& e) {
        set_error(e.what());
        return BDG_ERR_UNSTABLE;
    } catch (const std::exception& e) {
        set_error(e.what());
        return BDG_ERR_RUNTIME;
    } catch (...) {
        set_error("unknown exception");
        return BDG_ERR_RUNTIME;
    }
}

} // namespace bdg_detail
