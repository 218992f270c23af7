// Latency model of the owner-dataflow epoch: every hub row (item) belongs to one 16-lane group, which walks its
// tuples in sequence order with the hub row on chip; spoke rows (users) travel between groups through HBM with a version counter.
// One pass in sequence order gives the finish time of every tuple:
//   issue  = max(done of this owner's tuple D places back (prefetch depth), publish time of the spoke row's previous tuple)
//   start  = max(issue + L_load, done of this owner's previous tuple)
//   done   = start + t_same (same hub row as the owner's previous tuple) | t_switch (another hub row)
//   publish(spoke) = done + L_pub
// A spoke row whose previous tuple belongs to the same owner needs no publish (the owner's own earlier store; program order), and none
// at all when that was the owner's previous tuple (the row is still in registers).
// usage (through ctypes, tools/owner_sim.py): owner_sim(n, hub, spoke, n_hubs, n_spokes, groups, D, L_load, L_pub, t_same, t_switch, out[4])
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { int64_t load; int g; } HeapE;
static void sift(HeapE *h, int n, int i) {
    for (;;) {
        int l = 2 * i + 1, r = l + 1, m = i;
        if (l < n && h[l].load < h[m].load) m = l;
        if (r < n && h[r].load < h[m].load) m = r;
        if (m == i) return;
        HeapE t = h[i]; h[i] = h[m]; h[m] = t; i = m;
    }
}
static const int64_t *g_deg;
static int cmp_deg(const void *a, const void *b) {
    int64_t da = g_deg[*(const int *)a], db = g_deg[*(const int *)b];
    return da < db ? 1 : da > db ? -1 : 0;
}

int owner_sim(int64_t n, const int32_t *hub, const int32_t *spoke, int n_hubs, int n_spokes, int groups, int D, double L_load,
              double L_pub, double t_same, double t_switch, double *out) {
    int64_t *deg = calloc(n_hubs, 8);
    for (int64_t t = 0; t < n; ++t) deg[hub[t]]++;
    int *order = malloc(sizeof(int) * n_hubs);
    for (int i = 0; i < n_hubs; ++i) order[i] = i;
    g_deg = deg;
    qsort(order, n_hubs, sizeof(int), cmp_deg);
    HeapE *heap = malloc(sizeof(HeapE) * groups);
    for (int g = 0; g < groups; ++g) heap[g] = (HeapE){0, g};
    int *owner = malloc(sizeof(int) * n_hubs);
    for (int i = 0; i < n_hubs; ++i) { // longest-processing-time-first bin packing
        owner[order[i]] = heap[0].g;
        heap[0].load += deg[order[i]];
        sift(heap, groups, 0);
    }
    int64_t max_load = 0;
    for (int g = 0; g < groups; ++g) if (heap[g].load > max_load) max_load = heap[g].load;
    double *ring = calloc((size_t)groups * D, 8); // done times of the owner's last D tuples
    int64_t *cnt = calloc(groups, 8);
    int *last_hub = malloc(sizeof(int) * groups);
    memset(last_hub, 0xff, sizeof(int) * groups);
    double *pub = calloc(n_spokes, 8);
    int *pub_g = malloc(sizeof(int) * n_spokes);
    int64_t *pub_c = calloc(n_spokes, 8);
    memset(pub_g, 0xff, sizeof(int) * n_spokes);
    double hot_wait = 0;
    int hot_g = -1;
    double makespan = 0, wait_dep = 0;
    int64_t n_wait = 0;
    for (int64_t t = 0; t < n; ++t) {
        const int g = owner[hub[t]];
        const int64_t c = cnt[g];
        double *r = ring + (size_t)g * D;
        const double prev = c ? r[(c - 1) % D] : 0.0;
        const double back = c >= D ? r[c % D] : 0.0;
        if (hot_g < 0) hot_g = owner[order[0]];
        const int sp = spoke[t];
        double ready = pub[sp];
        double start;
        if (ready <= back) start = back + L_load;             // the speculative prefetch (row + version, D tuples ahead) was valid
        else start = (ready > prev ? ready : prev) + 0.5 + L_load; // slow path: flush, spin on the version, load
        if (pub_g[sp] == g && pub_c[sp] == c - 1) start = prev; // forwarded in registers
        if (start > prev + 1e-12) { wait_dep += start - prev; n_wait++; if (g == hot_g) hot_wait += start - prev; }
        if (start < prev) start = prev;
        const double done = start + (last_hub[g] == hub[t] ? t_same : t_switch);
        last_hub[g] = hub[t];
        r[c % D] = done;
        cnt[g] = c + 1;
        pub[sp] = done + L_pub;
        pub_g[sp] = g;
        pub_c[sp] = c;
        if (done > makespan) makespan = done;
    }
    out[0] = makespan;
    out[1] = (double)max_load;
    out[2] = wait_dep;
    out[3] = hot_wait;
    free(deg); free(order); free(heap); free(owner); free(ring); free(cnt); free(last_hub); free(pub); free(pub_g); free(pub_c);
    return 0;
}
