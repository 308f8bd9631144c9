// TEST INFRASTRUCTURE ONLY -- a stand-in for librccl.so that lets several ranks share ONE GPU.
//
// RCCL refuses two ranks on the same device, and the GPU boxes tests run on have one. The product binds
// RCCL with dlopen (blitzdg_amd/csrc/hip/sw2d_device.hip: RcclApi); with BDG_RCCL_LIBRARY pointing here,
// the multi-process path -- one process per rank, file rendezvous of the id, ncclCommInitRank, the grouped
// ncclSend / ncclRecv of every stage on the exchange stream, ncclAllReduce for dt / barriers -- runs
// unchanged, and only the transport underneath differs: messages travel through files in a shared
// directory, staged through the host, with the calling stream drained before a buffer is read and after
// it is written (a conservative rendering of NCCL's stream semantics; no performance meaning).
// Nothing in blitzdg_amd/ refers to this file.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <unistd.h>
#include <vector>

namespace {

struct Comm {
    std::string dir;
    int rank = 0, nranks = 1;
    std::vector<unsigned long long> sendSeq, recvSeq;
    unsigned long long arSeq = 0;
};

struct Op { bool send; void* buf; size_t bytes; int peer; Comm* comm; hipStream_t stream; };
thread_local int groupDepth = 0;
thread_local std::vector<Op> pending;

size_t typeSize(ncclDataType_t t) {
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    default: return 8;
    }
}

std::string baseDir() {
    const char* d = std::getenv("BDG_MOCK_RCCL_DIR");
    return d ? d : "/tmp";
}

bool readWhole(const std::string& path, std::vector<char>& data, size_t bytes) {
    std::ifstream in(path, std::ios::binary);
    if (!in) return false;
    data.resize(bytes);
    in.read(data.data(), static_cast<std::streamsize>(bytes));
    return static_cast<size_t>(in.gcount()) == bytes;
}

void writeAtomically(const std::string& path, const void* data, size_t bytes) {
    const std::string tmp = path + ".tmp" + std::to_string(getpid());
    {
        std::ofstream out(tmp, std::ios::binary);
        out.write(static_cast<const char*>(data), static_cast<std::streamsize>(bytes));
    }
    std::rename(tmp.c_str(), path.c_str());
}

ncclResult_t waitFor(const std::string& path, std::vector<char>& data, size_t bytes) {
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(120);
    while (!readWhole(path, data, bytes)) {
        if (std::chrono::steady_clock::now() > deadline) {
            std::fprintf(stderr, "mock rccl: timed out waiting for %s\n", path.c_str());
            return ncclSystemError;
        }
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    return ncclSuccess;
}

ncclResult_t execute(const Op& op) {
    Comm* c = op.comm;
    if (op.send) {
        if (hipStreamSynchronize(op.stream) != hipSuccess) return ncclUnhandledCudaError;
        std::vector<char> host(op.bytes);
        if (hipMemcpy(host.data(), op.buf, op.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
        const std::string path = c->dir + "/m_" + std::to_string(c->rank) + "_" + std::to_string(op.peer) + "_" +
                                 std::to_string(c->sendSeq[op.peer]++);
        writeAtomically(path, host.data(), op.bytes);
        return ncclSuccess;
    }
    const std::string path = c->dir + "/m_" + std::to_string(op.peer) + "_" + std::to_string(c->rank) + "_" +
                             std::to_string(c->recvSeq[op.peer]++);
    std::vector<char> host;
    const ncclResult_t r = waitFor(path, host, op.bytes);
    if (r != ncclSuccess) return r;
    if (hipStreamSynchronize(op.stream) != hipSuccess) return ncclUnhandledCudaError;
    if (hipMemcpy(op.buf, host.data(), op.bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    std::remove(path.c_str());
    return ncclSuccess;
}

ncclResult_t flush() {
    ncclResult_t rc = ncclSuccess;
    for (const Op& op : pending)   // all sends first: a send never waits for its receiver
        if (op.send && rc == ncclSuccess) rc = execute(op);
    for (const Op& op : pending)
        if (!op.send && rc == ncclSuccess) rc = execute(op);
    pending.clear();
    return rc;
}

} // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    std::memset(id, 0, sizeof(*id));
    std::snprintf(id->internal, sizeof(id->internal), "bdgmock_%d_%lld", getpid(),
                  static_cast<long long>(std::chrono::steady_clock::now().time_since_epoch().count()));
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    Comm* c = new Comm();
    c->rank = rank;
    c->nranks = nranks;
    c->sendSeq.assign(nranks, 0);
    c->recvSeq.assign(nranks, 0);
    c->dir = baseDir() + "/" + std::string(id.internal);
    const std::string cmd = "mkdir -p '" + c->dir + "'";
    if (std::system(cmd.c_str()) != 0) return ncclSystemError;
    *comm = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    delete reinterpret_cast<Comm*>(comm);
    return ncclSuccess;
}

ncclResult_t ncclGroupStart() { ++groupDepth; return ncclSuccess; }

ncclResult_t ncclGroupEnd() {
    if (--groupDepth > 0) return ncclSuccess;
    groupDepth = 0;
    return flush();
}

ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
    pending.push_back(Op{true, const_cast<void*>(buf), count * typeSize(type), peer, reinterpret_cast<Comm*>(comm), stream});
    return groupDepth > 0 ? ncclSuccess : flush();
}

ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
    pending.push_back(Op{false, buf, count * typeSize(type), peer, reinterpret_cast<Comm*>(comm), stream});
    return groupDepth > 0 ? ncclSuccess : flush();
}

ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t type, ncclRedOp_t op,
                           ncclComm_t comm, hipStream_t stream) {
    if (type != ncclDouble) return ncclInvalidArgument;
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    std::vector<double> mine(count), acc(count);
    if (hipMemcpy(mine.data(), sendbuff, count * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    const unsigned long long seq = c->arSeq++;
    writeAtomically(c->dir + "/ar_" + std::to_string(seq) + "_" + std::to_string(c->rank), mine.data(), count * sizeof(double));
    for (int r = 0; r < c->nranks; ++r) {
        std::vector<char> raw;
        const ncclResult_t rc = waitFor(c->dir + "/ar_" + std::to_string(seq) + "_" + std::to_string(r), raw, count * sizeof(double));
        if (rc != ncclSuccess) return rc;
        const double* v = reinterpret_cast<const double*>(raw.data());
        for (size_t i = 0; i < count; ++i) {
            if (r == 0) acc[i] = v[i];
            else if (op == ncclSum) acc[i] += v[i];
            else if (op == ncclMax) acc[i] = v[i] > acc[i] ? v[i] : acc[i];
            else if (op == ncclMin) acc[i] = v[i] < acc[i] ? v[i] : acc[i];
            else return ncclInvalidArgument;
        }
    }
    if (hipMemcpy(recvbuff, acc.data(), count * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "mock rccl error"; }

} // extern "C"
