// owner_gate.hpp -- who may launch a persistent owner epoch on a device, and when: the per-device gate of this process and the advisory
// file lock between processes (used by cmi_api.cpp enqueue_levels; internal header).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <thread>

// Owner epochs are persistent launches: every workgroup must be resident, so the workgroups of the epochs in flight on one device must
// fit it together.  Inside a process a per-device GATE counts them (capacity = the device's compute units, one owner workgroup each):
// an instance with cmi_set_device_share(F) launches ~1 / F of the device and F such epochs run side by side (`cv -p on`); an instance
// without the hint takes the whole device and is alone.  Across PROCESSES an advisory flock() on a per-device file (held by a process
// while any of its owner epochs is in flight) keeps two processes' persistent kernels apart.  (This is about RESIDENCY only.  Round 5
// also kept the team form away from other owner epochs because it was measured inexact beside them; the cause was a store-data hazard
// in the record stores, fixed in owner_kernels.hip owner_st_words, and those rules are gone: docs/history/r06.md 1.)
//  * The file lock is polled WITHOUT the gate's mutex (a 60-s poll under the mutex blocked every other fold of the
//    process); one thread acquires for the process, the others wait on the condition variable.
//  * Fair hand-over: a process whose folds overlap their epochs never sees `holders` reach 0 by itself, and a lone fast process
//    re-acquires microseconds after releasing.  A process that WAITS for the lock holds a shared flock on a second file (".waiters")
//    while it polls; a holder that has kept the lock for 50 ms tests that file (one non-blocking exclusive attempt) and, if somebody
//    waits, DRAINS: no new epoch is admitted, the last one out releases the lock, and the process stays away from it for 2 ms -- the
//    waiting peer polls every 250 us.  Nobody waiting: no pause, nothing changes.
//  * The lock file cannot be opened (read-only $TMPDIR ...): the epoch is refused (CMI_E_BUSY) instead of launched unprotected, unless
//    CMI_OWNER_NO_LOCK=1 says this process is the device's only user.  The lock directory is per uid (0700), so processes of DIFFERENT
//    users are not serialised against each other: a GPU shared across uids needs one process per GPU.
#include <fcntl.h>
#include <sys/file.h>
#include <sys/stat.h>
#include <unistd.h>
struct OwnerDeviceGate {
    static constexpr int MAX_DEV = 64;
    std::mutex m;
    std::condition_variable cv;
    int in_use = 0;          // workgroups of the owner epochs in flight
    int holders = 0;         // epochs in flight (the file lock is held while > 0)
    int fd = -1, fd_wait = -1; // the lock file and the waiters' file, opened once per process and device
    bool locked = false;     // this process holds the flock
    bool acquiring = false;  // one thread is polling the flock (without `m`)
    bool draining = false;   // fairness: no admissions until the epochs in flight are done and the flock has been released
    std::chrono::steady_clock::time_point since, not_before; // when the flock was taken; earliest re-acquisition after a fair release
    static OwnerDeviceGate &of(int dev) {
        static OwnerDeviceGate g[MAX_DEV];
        return g[dev >= 0 && dev < MAX_DEV ? dev : 0];
    }
};
struct OwnerDeviceLock {
    OwnerDeviceGate &g;
    int wgs;
    bool ok = true;  // false: the epoch must not be launched (`why` says which of the two reasons)
    const char *why = "";
    OwnerDeviceLock(int dev, int workgroups, int capacity) : g(OwnerDeviceGate::of(dev)), wgs(std::max(1, workgroups)) {
        static const bool no_lock = getenv("CMI_OWNER_NO_LOCK") != nullptr;
        std::unique_lock<std::mutex> lk(g.m);
        while (true) {
            g.cv.wait(lk, [&] { return !g.acquiring && !g.draining && (g.in_use == 0 || g.in_use + wgs <= capacity); });
            if (g.locked || no_lock) break;
            // this thread takes the flock for the process; the mutex is NOT held while it polls
            g.acquiring = true;
            const auto not_before = g.not_before;
            lk.unlock();
            bool got = false, opened = true;
            if (g.fd < 0) { // (only the acquiring thread touches fd)
                char bus[64] = "";
                if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, dev) != hipSuccess) snprintf(bus, sizeof bus, "dev%d", dev);
                for (char *c = bus; *c; ++c)
                    if (*c == ':' || *c == '/' || *c == '.') *c = '_';
                // a per-uid directory (0700) under $TMPDIR, so the lock file is neither world-writable nor at a path another user can
                // plant a symlink on; O_NOFOLLOW refuses a planted link anyway, O_CLOEXEC keeps the descriptor out of forked children
                const char *tmp = getenv("CMI_OWNER_LOCK_DIR"); // (where the lock files live, if not under $TMPDIR: every process that shares the GPU must agree)
                if (!tmp || !*tmp) tmp = getenv("TMPDIR");
                char dir[200], path[300];
                snprintf(dir, sizeof dir, "%s/cmi_locks_%u", tmp && *tmp ? tmp : "/tmp", (unsigned)getuid());
                (void)mkdir(dir, 0700);
                snprintf(path, sizeof path, "%s/owner_epoch_%s.lock", dir, bus);
                g.fd = open(path, O_CREAT | O_RDWR | O_NOFOLLOW | O_CLOEXEC, 0600);
                opened = g.fd >= 0;
                if (opened) {
                    snprintf(path, sizeof path, "%s/owner_epoch_%s.waiters", dir, bus);
                    g.fd_wait = open(path, O_CREAT | O_RDWR | O_NOFOLLOW | O_CLOEXEC, 0600); // (best effort: without it nobody sees us wait)
                }
            }
            if (opened) {
                std::this_thread::sleep_until(not_before); // (a fair release just happened: let the waiting peer in first)
                // bounded wait: a stopped or hung peer holding the lock must not block this process for ever; an owner epoch lasts well
                // under a second.  After 60 s the epoch is NOT launched beside the other process's persistent kernel (that could stall both)
                const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(60);
                bool announced = false;
                while (!(got = flock(g.fd, LOCK_EX | LOCK_NB) == 0) && std::chrono::steady_clock::now() < deadline) {
                    if (!announced && g.fd_wait >= 0) announced = flock(g.fd_wait, LOCK_SH | LOCK_NB) == 0; // "somebody waits"
                    usleep(250);
                }
                if (announced) flock(g.fd_wait, LOCK_UN);
            }
            lk.lock();
            g.acquiring = false;
            if (!got) {
                ok = false;
                why = opened ? "another process has held the owner-epoch lock of the device for 60 s"
                             : "the owner-epoch lock file under $CMI_OWNER_LOCK_DIR or $TMPDIR (/cmi_locks_<uid>/) cannot be opened (set CMI_OWNER_NO_LOCK=1 if this "
                               "process is the only user of the GPU)";
                g.cv.notify_all();
                return; // (nothing taken: in_use / holders unchanged)
            }
            g.locked = true;
            g.since = std::chrono::steady_clock::now();
            g.cv.notify_all();
            // (loop: the capacity predicate is re-evaluated under the mutex)
        }
        g.in_use += wgs;
        ++g.holders;
    }
    ~OwnerDeviceLock() {
        if (!ok) return;
        std::lock_guard<std::mutex> lk(g.m);
        g.in_use -= wgs;
        --g.holders;
        if (g.locked && !g.draining && g.fd_wait >= 0 && std::chrono::steady_clock::now() - g.since > std::chrono::milliseconds(50)) {
            if (flock(g.fd_wait, LOCK_EX | LOCK_NB) == 0) { // nobody holds the shared lock: nobody waits
                flock(g.fd_wait, LOCK_UN);
                g.since = std::chrono::steady_clock::now();
            } else g.draining = true;
        }
        if (g.holders == 0 && g.locked) {
            flock(g.fd, LOCK_UN);
            g.locked = false;
            if (g.draining) g.not_before = std::chrono::steady_clock::now() + std::chrono::milliseconds(2);
            g.draining = false;
        }
        g.cv.notify_all();
    }
};
