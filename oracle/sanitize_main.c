/* ASan/UBSan driver for the CPU oracle (test infrastructure): World::init restatement incl. water fill and edits,
 * then an image and a ray list through chunkmarch.  Exit code 0 and no sanitizer report = clean. */
#include "svo_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

int main(void)
{
    orc_world w;
    orc_terrain tp = { 6, 0, 64.0f, 16.0f, 3, 1, 6.0f, 6 };
    int ccm[3] = { -1, 0, -1 };
    orc_world_init(&w, 2, 1, 2, 128, ccm, &tp);
    orc_delta dt, dw;
    orc_vec3 lo = { -100, 40, -100 }, hi = { -60, 90, -50 };
    orc_build(&w.chunk[orc_world_index3(&w, -1, 0, -1)], lo, hi, 5, &dt, &dw);
    orc_vec3 lo2 = { -128, 0, -128 }, hi2 = { 0, 30, -90 };
    orc_destroy(&w.chunk[orc_world_index3(&w, -1, 0, -1)], lo2, hi2, &dt, &dw);

    orc_camera cam = { { 0, 150, -170 }, { 0, -0.5f, 0.8660254f }, { -1, 0, 0 }, { 0, 0.8660254f, 0.5f }, 1.0264f, 0.57735f, 96, 54 };
    orc_params prm = { 0, 0, 0, 0, 1, { 1, -1, 0 } };
    orc_hit *out = (orc_hit *)malloc(sizeof(orc_hit) * 96 * 54);
    orc_counters *cnt = (orc_counters *)malloc(sizeof(orc_counters) * 96 * 54);
    unsigned long long rays = orc_trace_image(&w, &cam, &prm, 0, 0, 96, 54, out, cnt, 3);
    unsigned hits = 0;
    for (int i = 0; i < 96 * 54; ++i) hits += out[i].flags & 1;

    /* axis-parallel / NaN / zero directions and origins on lattice planes */
    float o[8][3] = { { 0, 0, 0 }, { -128, 64, -128 }, { 64, 500, 64 }, { 10, 100, 10 }, { 10, 100, 10 }, { 0, 128, 0 }, { -64, 20, -64 }, { 127.99f, 1, 127.99f } };
    float d[8][3] = { { 0, 0, 1 }, { 1, 0, 0 }, { 0, -1, 0 }, { 0, 0, 0 }, { NAN, 1, 0 }, { 0, -1, 0 }, { 0.57735f, 0.57735f, 0.57735f }, { -1, 0, 0 } };
    orc_hit h8[8];
    rays += orc_trace_rays(&w, &o[0][0], &d[0][0], 8, &prm, h8, NULL, 1);
    printf("rays %llu hits %u\n", rays, hits);
    free(out); free(cnt);
    orc_world_deinit(&w);
    return hits > 100 ? 0 : 1;
}
