// Drives tests/stub_rccl between two forked processes (all-reduce + a grouped send/recv ring), for debugging the stand-in itself.
#include "../tests/stub_rccl/ps_stub_rccl.hip"
#include <sys/wait.h>
int main() {
    StubUid id; ncclGetUniqueId(&id);
    const pid_t pid = fork();
    const int rank = pid ? 0 : 1, world = 2;
    void* comm = nullptr;
    int rc = ncclCommInitRank(&comm, world, id, rank);
    std::printf("[%d] init rc %d\n", rank, rc); std::fflush(stdout);
    if (rc) return 1;
    hipStream_t st; { int lo = 0, hi = 0; hipDeviceGetStreamPriorityRange(&lo, &hi); hipError_t e = hipStreamCreateWithPriority(&st, hipStreamNonBlocking, getenv("T_PRIO") ? hi : 0); std::printf("[%d] stream prio range %d..%d -> %s\n", rank, lo, hi, hipGetErrorString(e)); }
    double* buf; hipMalloc((void**)&buf, 1024); double* rbuf; hipMalloc((void**)&rbuf, 1024);
    for (int it = 0; it < 300; ++it) {
        double h[4] = {1.0 + rank, 2.0, 3.0 * rank, (double)it};
        hipMemcpyAsync(buf, h, 32, hipMemcpyHostToDevice, st);
        rc = ncclAllReduce(buf, buf, 4, 8, 0, comm, st);
        if (rc) { std::printf("[%d] allreduce %d rc %d\n", rank, it, rc); return 2; }
        ncclGroupStart();
        ncclSend(buf, 4, 8, 1 - rank, comm, st);
        ncclRecv(rbuf, 4, 8, 1 - rank, comm, st);
        rc = ncclGroupEnd();
        if (rc) { std::printf("[%d] sendrecv %d rc %d\n", rank, it, rc); return 3; }
        double o[4], r[4];
        hipMemcpyAsync(o, buf, 32, hipMemcpyDeviceToHost, st); hipMemcpyAsync(r, rbuf, 32, hipMemcpyDeviceToHost, st);
        hipStreamSynchronize(st);
        if (it % 100 == 0 || o[0] != 3. || r[3] != 2. * it) std::printf("[%d] it %d allreduce %g %g %g %g   recv %g %g %g %g\n", rank, it, o[0], o[1], o[2], o[3], r[0], r[1], r[2], r[3]); std::fflush(stdout);
    }
    ncclCommDestroy(comm);
    if (pid) { int s = 0; waitpid(pid, &s, 0); std::printf("child exit %d\n", WEXITSTATUS(s)); }
    return 0;
}
