// Probe: which cross-process, stream-ordered synchronisation primitives work between two processes on ONE GPU of this pool?
//  (1) hipIpcEventHandle events: record in A, hipStreamWaitEvent / hipEventQuery / hipEventSynchronize in B
//  (2) hipStreamWriteValue32 (A) / hipStreamWaitValue32 (B) on hipIpcMemHandle-shared device memory
//  (3) a spin kernel in B on a flag in IPC-shared memory that a kernel in A sets
// build: hipcc -O2 --offload-arch=gfx950 scripts/t_ipc_event.hip -o scripts/_bin/t_ipc_event
#include <hip/hip_runtime.h>
#include <unistd.h>
#include <sys/wait.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <ctime>
#define CK(x) do { hipError_t e = (x); std::printf("[%s] %-70s -> %s\n", who, #x, hipGetErrorString(e)); std::fflush(stdout); } while (0)
__global__ void k_set(volatile int* f, int v) { __threadfence_system(); *f = v; __threadfence_system(); }
__global__ void k_spin(volatile int* f, int v, int* out) {
    long long n = 0;
    while (*f < v && n < 40000000ll) ++n;
    *out = (*f >= v) ? 1 : -1;
}
struct Msg { hipIpcEventHandle_t ev; hipIpcMemHandle_t mem; };
int main() {
    int p2c[2], c2p[2];
    if (pipe(p2c) || pipe(c2p)) return 1;
    const pid_t pid = fork();                       // before any HIP call
    const char* who = pid ? "A" : "B";
    if (pid) {   // A: producer
        hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming | hipEventInterprocess));
        int* flag; CK(hipMalloc((void**)&flag, 4096)); CK(hipMemset(flag, 0, 4096));
        Msg m; CK(hipIpcGetEventHandle(&m.ev, ev)); CK(hipIpcGetMemHandle(&m.mem, flag));
        CK(hipEventRecord(ev, st));                  // a first record before the handle travels
        CK(hipStreamSynchronize(st));
        if (write(p2c[1], &m, sizeof(m)) != (ssize_t)sizeof(m)) return 2;
        char c; if (read(c2p[0], &c, 1) != 1) return 3;   // B has opened
        // (1) record after some work
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(1), 0, st, (volatile int*)(flag + 128), 1, flag + 200);   // ~0.3 s of busy stream: B waits on an event that is NOT ready
        hipLaunchKernelGGL(k_set, dim3(1), dim3(1), 0, st, (volatile int*)(flag + 16), 7);
        CK(hipEventRecord(ev, st));
        if (write(p2c[1], "r", 1) != 1) return 4;          // "record enqueued"
        if (read(c2p[0], &c, 1) != 1) return 5;            // B done with (1)
        // (2) write value
        if (write(p2c[1], "w", 1) != 1) return 6;            // B enqueues its wait FIRST
        usleep(800000);
        hipLaunchKernelGGL(k_set, dim3(1), dim3(1), 0, st, (volatile int*)(flag + 48), 42);
        CK(hipStreamWriteValue32(st, flag + 32, 5, 0));
        CK(hipStreamSynchronize(st));
        if (read(c2p[0], &c, 1) != 1) return 7;
        // (3) kernel sets flag
        hipLaunchKernelGGL(k_set, dim3(1), dim3(1), 0, st, (volatile int*)(flag + 64), 9);
        CK(hipStreamSynchronize(st));
        if (write(p2c[1], "k", 1) != 1) return 8;
        if (read(c2p[0], &c, 1) != 1) return 9;
        int status = 0; waitpid(pid, &status, 0);
        std::printf("[A] child exit %d\n", WEXITSTATUS(status));
        return 0;
    }
    // B: consumer
    Msg m; if (read(p2c[0], &m, sizeof(m)) != (ssize_t)sizeof(m)) return 2;
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t ev = nullptr; CK(hipIpcOpenEventHandle(&ev, m.ev));
    int* flag = nullptr; CK(hipIpcOpenMemHandle((void**)&flag, m.mem, hipIpcMemLazyEnablePeerAccess));
    int* out; CK(hipMalloc((void**)&out, 64)); CK(hipMemset(out, 0, 64));
    if (write(c2p[1], "o", 1) != 1) return 3;
    char c; if (read(p2c[0], &c, 1) != 1) return 4;
    CK(hipEventQuery(ev));
    CK(hipStreamWaitEvent(st, ev, 0));
    CK(hipStreamWaitEvent(nullptr, ev, 0));
    CK(hipEventSynchronize(ev));
    int h = 0; CK(hipMemcpyAsync(&h, flag + 16, 4, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
    std::printf("[B] (1) value behind the event: %d (expect 7)\n", h);
    if (write(c2p[1], "1", 1) != 1) return 5;
    if (read(p2c[0], &c, 1) != 1) return 6;
    {
        timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0);
        CK(hipStreamWaitValue32(st, flag + 32, 5, hipStreamWaitValueGte, 0xffffffffu));
        CK(hipMemcpyAsync(&h, flag + 48, 4, hipMemcpyDeviceToHost, st));
        clock_gettime(CLOCK_MONOTONIC, &t1);
        std::printf("[B] (2) enqueue took %.3f s\n", (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec));
        CK(hipStreamSynchronize(st));
        clock_gettime(CLOCK_MONOTONIC, &t1);
        std::printf("[B] (2) data behind the value wait: %d (expect 42), waited %.3f s (expect ~0.8)\n", h, (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec));
    }
    if (write(c2p[1], "2", 1) != 1) return 7;
    if (read(p2c[0], &c, 1) != 1) return 8;
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(1), 0, st, (volatile int*)(flag + 64), 9, out);
    CK(hipMemcpyAsync(&h, out, 4, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
    std::printf("[B] (3) spin kernel result: %d (expect 1)\n", h);
    if (write(c2p[1], "3", 1) != 1) return 9;
    return 0;
}
