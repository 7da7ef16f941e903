/* A plain-C consumer of the batched device-task C ABI (include/motcpp_amd.h): what a cgo / JNI / ctypes binding of the
 * reference would call. Builds N small IoU-cost assignment problems from boxes, solves them in ONE launch with the
 * costs computed inside the solver, and checks (a) the result against mot_lap_geom_host problem by problem, (b) that
 * the launch flags do what the header says: MOT_LAP_F_PLAIN with a MOT_COST_BOTSORT task is refused (info = -1,
 * everything unmatched) rather than solved as something else. Compiled as C (the header must be valid C); needs an
 * MI355X at run time. */
#include <motcpp_amd.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(c) do { if (!(c)) { printf("FAILED %s:%d: %s (%s)\n", __FILE__, __LINE__, #c, ctx ? mot_ctx_last_error(ctx) : ""); return 1; } } while (0)
#define OK(call) CHECK((call) == MOT_OK)

enum { P = 5, N = 40, M = 24 };

static float frand(unsigned* s) { *s = *s * 1664525u + 1013904223u; return (float)((*s >> 8) & 0xffff) / 65535.0f; }

int main(void) {
  mot_ctx* ctx = NULL;
  OK(mot_ctx_create(0, NULL, &ctx));
  /* boxes: planes [4][P*N] rows, [4][P*M] columns, confidences [P*M] */
  static float a[4 * P * N], b[4 * P * M], conf[P * M];
  unsigned seed = 7;
  for (int p = 0; p < P; ++p) {
    for (int i = 0; i < N; ++i) {
      const float cx = 600.f * frand(&seed), cy = 400.f * frand(&seed), w = 30.f + 40.f * frand(&seed), h = 60.f + 60.f * frand(&seed);
      const int k = p * N + i;
      a[k] = cx - w / 2; a[P * N + k] = cy - h / 2; a[2 * P * N + k] = cx + w / 2; a[3 * P * N + k] = cy + h / 2;
    }
    for (int j = 0; j < M; ++j) {  /* a detection near row j, a little off */
      const int k = p * M + j, r = p * N + j;
      for (int c = 0; c < 4; ++c) b[c * P * M + k] = a[c * P * N + r] + 4.f * (frand(&seed) - 0.5f);
      conf[k] = 0.5f + 0.5f * frand(&seed);
    }
  }
  float *da, *db, *dconf;
  int32_t *dx, *dy, *dinfo;
  void* dwork[P];
  mot_lap_task tasks[P], *dtasks;
  OK(mot_malloc(ctx, sizeof a, (void**)&da)); OK(mot_malloc(ctx, sizeof b, (void**)&db)); OK(mot_malloc(ctx, sizeof conf, (void**)&dconf));
  OK(mot_malloc(ctx, sizeof(int32_t) * P * N, (void**)&dx)); OK(mot_malloc(ctx, sizeof(int32_t) * P * M, (void**)&dy));
  OK(mot_malloc(ctx, sizeof(int32_t) * P, (void**)&dinfo)); OK(mot_malloc(ctx, sizeof tasks, (void**)&dtasks));
  OK(mot_memcpy_h2d(ctx, da, a, sizeof a)); OK(mot_memcpy_h2d(ctx, db, b, sizeof b)); OK(mot_memcpy_h2d(ctx, dconf, conf, sizeof conf));
  memset(tasks, 0, sizeof tasks);
  for (int p = 0; p < P; ++p) {
    OK(mot_malloc(ctx, mot_lap_work_bytes(N, M), &dwork[p]));
    mot_lap_task* t = &tasks[p];
    t->n = N; t->m = M; t->thresh = 0.8f; t->x = dx + p * N; t->y = dy + p * M; t->mode = MOT_LAP_PLAIN; t->info = dinfo + p; t->work = dwork[p];
    t->geom.n = N; t->geom.m = M; t->geom.a = da + p * N; t->geom.lda = P * N; t->geom.b = db + p * M; t->geom.ldb = P * M;
    t->geom.bconf = dconf + p * M; t->geom.mode = MOT_COST_IOU_DIST_FUSE;
  }
  static int32_t x[P * N], y[P * M], info[P];
  for (int round = 0; round < 2; ++round) {  /* with and without the plain-cost promise: same answers */
    OK(mot_memcpy_h2d(ctx, dtasks, tasks, sizeof tasks));
    OK(mot_lap_solve(ctx, dtasks, P, N, M, MOT_LAP_F_GEOM | (round ? MOT_LAP_F_PLAIN : 0)));
    OK(mot_memcpy_d2h(ctx, x, dx, sizeof x)); OK(mot_memcpy_d2h(ctx, y, dy, sizeof y)); OK(mot_memcpy_d2h(ctx, info, dinfo, sizeof info));
    OK(mot_ctx_sync(ctx));
    for (int p = 0; p < P; ++p) {
      float ar[N * 4], br[M * 4];  /* the host convenience entry point takes row-major n x 4 */
      for (int i = 0; i < N; ++i) for (int c = 0; c < 4; ++c) ar[i * 4 + c] = a[c * P * N + p * N + i];
      for (int j = 0; j < M; ++j) for (int c = 0; c < 4; ++c) br[j * 4 + c] = b[c * P * M + p * M + j];
      int rx[N], ry[M], rinfo = 0, matched = 0;
      OK(mot_lap_geom_host(ctx, ar, N, br, M, conf + p * M, MOT_COST_IOU_DIST_FUSE, 0.8f, MOT_LAP_PLAIN, 0.f, rx, ry, NULL, &rinfo, NULL));
      CHECK(info[p] == rinfo);
      for (int i = 0; i < N; ++i) { CHECK(x[p * N + i] == rx[i]); matched += rx[i] >= 0; }
      for (int j = 0; j < M; ++j) CHECK(y[p * M + j] == ry[j]);
      CHECK(matched >= M / 2);  /* the detections were placed on the first M rows */
    }
  }
  /* a false promise is refused */
  tasks[2].geom.mode = MOT_COST_BOTSORT; tasks[2].geom.lde = -1; tasks[2].geom.prox_thresh = 0.5f; tasks[2].geom.app_thresh = 0.25f;
  OK(mot_memcpy_h2d(ctx, dtasks, tasks, sizeof tasks));
  OK(mot_lap_solve(ctx, dtasks, P, N, M, MOT_LAP_F_GEOM | MOT_LAP_F_PLAIN));
  OK(mot_memcpy_d2h(ctx, x, dx, sizeof x)); OK(mot_memcpy_d2h(ctx, info, dinfo, sizeof info));
  OK(mot_ctx_sync(ctx));
  CHECK(info[2] == -1 && info[1] != -1);
  for (int i = 0; i < N; ++i) CHECK(x[2 * N + i] == -1);
  /* ... and without the promise the same task is solved */
  OK(mot_lap_solve(ctx, dtasks, P, N, M, MOT_LAP_F_GEOM));
  OK(mot_memcpy_d2h(ctx, info, dinfo, sizeof info));
  OK(mot_ctx_sync(ctx));
  CHECK(info[2] == 0 || info[2] == 1);
  for (int p = 0; p < P; ++p) OK(mot_free(ctx, dwork[p]));
  OK(mot_free(ctx, da)); OK(mot_free(ctx, db)); OK(mot_free(ctx, dconf)); OK(mot_free(ctx, dx)); OK(mot_free(ctx, dy)); OK(mot_free(ctx, dinfo)); OK(mot_free(ctx, dtasks));
  OK(mot_ctx_destroy(ctx));
  printf("batched C ABI ok\n");
  return 0;
}
