/* Pooling statistics of the left-to-right isotonic sweep of one column (the stack the unimodal kernel keeps): for
 * tools/uni_depth.py.  gcc -O2 -shared -fPIC tools/uni_depth.c -o /tmp/uni_depth.so */
#include <stdlib.h>
void uni_depth(const float *y, long n, long stride, int ring_cap, int *out /* max depth, refills, spills, merges, max merges */,
               unsigned char *merges_per_step /* n entries or NULL */) {
    double *sy = malloc(sizeof(double) * (n + 1));
    double *cw = malloc(sizeof(double) * (n + 1));
    long top = 0;                /* entries on the stack (below the block being built) */
    long ring = 0, mem = 0;      /* entries in the ring (excluding the cached top) and spilled */
    int has_top = 0;
    long maxd = 0, refills = 0, spills = 0, merges = 0, maxm = 0;
    for (long i = 0; i < n; ++i) {
        double csy = y[i * stride], ccw = 1.0;
        if (i > 0) {
            if (has_top) {
                if (ring == ring_cap) { mem++; ring = ring_cap - 1; spills++; }
                ring++;
            }
            has_top = 1;
        }
        long m = 0;
        while (has_top && top > 0 && csy * cw[top - 1] <= sy[top - 1] * ccw) {
            csy += sy[top - 1]; ccw += cw[top - 1]; top--; m++;
            if (ring == 0) {
                if (mem == 0) { has_top = 0; continue; }
                long k = mem < ring_cap / 2 ? mem : ring_cap / 2;
                mem -= k; ring = k; refills++;
            }
            ring--;
        }
        sy[top] = csy; cw[top] = ccw; top++;
        if (top > maxd) maxd = top;
        merges += m; if (m > maxm) maxm = m;
        if (merges_per_step) merges_per_step[i] = (unsigned char)(m > 255 ? 255 : m);
    }
    out[0] = (int)maxd; out[1] = (int)refills; out[2] = (int)spills; out[3] = (int)merges; out[4] = (int)maxm;
    free(sy); free(cw);
}
