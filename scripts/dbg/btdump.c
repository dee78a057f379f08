// debugging aid: native backtraces of every thread of this process on stderr (scripts/dbg_threads.sh; PS_DBG_BT=<seconds> in tests/mp_rank.py)
// gcc -O1 -g -shared -fPIC -o libbtdump.so btdump.c -lpthread
#define _GNU_SOURCE
#include <dirent.h>
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/syscall.h>
#include <unistd.h>
static void handler(int sig) {
    (void)sig;
    void* fr[64];
    const int n = backtrace(fr, 64);
    char head[64];
    const int m = snprintf(head, sizeof head, "--- thread %ld\n", (long)syscall(SYS_gettid));
    if (write(2, head, (size_t)m) < 0) return;
    backtrace_symbols_fd(fr, n, 2);
}
void bt_dump_all(void) {
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = handler;
    sigaction(SIGUSR2, &sa, NULL);
    DIR* d = opendir("/proc/self/task");
    if (!d) return;
    struct dirent* e;
    while ((e = readdir(d))) {
        const long tid = atol(e->d_name);
        if (tid > 0) { syscall(SYS_tgkill, getpid(), tid, SIGUSR2); usleep(20000); }
    }
    closedir(d);
}
