/*
 * avk_cpus.h — how many host threads are worth starting: the logical CPUs the process may run on AND the CPU time its cgroup grants.
 * std::thread::hardware_concurrency() reports the former only; in a container with a CFS quota (the GPU boxes of this pool: 256 logical CPUs
 * visible, cpu.max "1600000 100000" = 16 CPUs) pools sized by it use the period's quota up in a fraction of the period and every thread of the
 * process — the one that submits GPU work included — is then stopped until the next period (cpu.stat nr_throttled, throttled_usec).
 */
#ifndef AVK_CPUS_H
#define AVK_CPUS_H
#include <sched.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

static inline unsigned avk_usable_cpus() {
    static const unsigned cached = [] {
        unsigned n = std::thread::hardware_concurrency();
        if (n < 1) n = 1;
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) {
            const unsigned a = (unsigned)CPU_COUNT(&set);
            if (a && a < n) n = a;
        }
        unsigned long long quota = 0, period = 0;
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) { /* cgroup v2: "<quota|max> <period>" */
            char q[32];
            if (fscanf(f, "%31s %llu", q, &period) == 2 && strcmp(q, "max") != 0) quota = strtoull(q, nullptr, 10);
            fclose(f);
        } else { /* cgroup v1 */
            long long q1 = -1;
            if (FILE *fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
                if (fscanf(fq, "%lld", &q1) != 1) q1 = -1;
                fclose(fq);
            }
            if (FILE *fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
                if (fscanf(fp, "%llu", &period) != 1) period = 0;
                fclose(fp);
            }
            if (q1 > 0) quota = (unsigned long long)q1;
        }
        if (quota && period) {
            const unsigned long long c = (quota + period - 1) / period;
            if (c && c < n) n = (unsigned)c;
        }
        if (const char *e = getenv("AVK_CPUS")) { /* an explicit count wins */
            const long v = strtol(e, nullptr, 10);
            if (v >= 1) n = (unsigned)v;
        }
        return n;
    }();
    return cached;
}
#endif
