/* examples/hf_interpolate_clip.c -- a multi-GPU host written against the C ABI only (include/hopperflow.h): converts a raw
 * NV12 / P010 clip from 23.976 fps to the target rate, frames entering and leaving through HOST memory, one worker PROCESS per
 * GPU, no exchange between the workers.
 *
 *   gcc -std=c11 -D_GNU_SOURCE -Iinclude examples/hf_interpolate_clip.c -Lhopperrender_amd/lib -lhopperflow \
 *       -Wl,-rpath,$PWD/hopperrender_amd/lib -o hf_interpolate_clip
 *   ./hf_interpolate_clip in.nv12 out.nv12 WIDTH HEIGHT HDR(0|1) TARGET_FPS GPUS [RADIUS] [SCENE_THRESHOLD]
 *
 * The parent sizes the output file and forks the workers BEFORE anything touches a GPU.  Worker r plans its chunk of the
 * timeline (hf_shard_timeline: 3 + 12 warm-up frames rebuild ring, previous flow and scene-change history), creates an
 * asynchronous context on GPU r % hf_device_count() and lets hf_hostio_run stream it: `fill` preads a source frame straight
 * into a page-locked buffer, `sink` pwrites an output frame at its FINAL offset of the output file -- the results are gathered
 * in index order by construction.  Same bytes as the sequential, blocking filter protocol (tests/test_hostio_gpu.py). */
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include "hopperflow.h"

typedef struct {
    int fd_in, fd_out;
    size_t frame_bytes;
    int64_t first_output;
    int copies;
} Io;

static int fill(void* user, int64_t k, void* pinned) {
    Io* io = (Io*)user;
    return pread(io->fd_in, pinned, io->frame_bytes, (off_t)k * (off_t)io->frame_bytes) == (ssize_t)io->frame_bytes ? 0 : 1;
}

static int sink(void* user, int64_t i, const void* frame, int32_t kind) {
    Io* io = (Io*)user;
    io->copies += kind == 0;
    return pwrite(io->fd_out, frame, io->frame_bytes, (off_t)(io->first_output + i) * (off_t)io->frame_bytes) == (ssize_t)io->frame_bytes ? 0 : 1;
}

static int worker(const char* in, const char* out, int W, int H, int hdr, int64_t src_t, int64_t tgt_t, int64_t n_frames, int world, int rank,
                  int radius, int threshold) {
    hf_timeline_chunk ch;
    if (hf_shard_timeline(n_frames, world, rank, src_t, tgt_t, 3, 12, &ch, NULL, NULL, 0)) { fprintf(stderr, "%s\n", hf_hostio_last_error(NULL)); return 1; }
    if (ch.n_periods == 0) return 0;
    /* everything that can fail without a GPU comes first; one exit path releases whatever exists */
    int rc = 1;
    int32_t* n_out = NULL;
    float* t = NULL;
    hf_ctx* ctx = NULL;
    hf_hostio* hio = NULL;
    Io io = {open(in, O_RDONLY), open(out, O_WRONLY), (size_t)W * (size_t)H * 3 / 2 * (hdr ? 2 : 1), ch.first_output, 0};
    if (io.fd_in < 0 || io.fd_out < 0) { perror("open"); goto done; }
    n_out = (int32_t*)malloc(sizeof(int32_t) * (size_t)ch.n_periods);
    t = (float*)malloc(sizeof(float) * (size_t)(ch.n_outputs ? ch.n_outputs : 1));
    if (!n_out || !t) { perror("malloc"); goto done; }
    if (hf_shard_timeline(n_frames, world, rank, src_t, tgt_t, 3, 12, &ch, n_out, t, ch.n_outputs)) { fprintf(stderr, "%s\n", hf_hostio_last_error(NULL)); goto done; }

    hf_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.struct_size = sizeof(cfg);
    cfg.is_hdr = hdr; cfg.frame_height = H; cfg.frame_width = W;
    cfg.delta_scalar = 8; cfg.neighbor_scalar = 6; cfg.black_level = 0.0f; cfg.white_level = 255.0f; cfg.max_calc_res = 270;
    cfg.device_index = rank % (hf_device_count() > 0 ? hf_device_count() : 1);
    cfg.search_radius = radius;
    cfg.flags = HF_FLAG_ASYNC | HF_FLAG_DUAL_STREAM;
    if (hf_create(&cfg, &ctx)) { fprintf(stderr, "rank %d: %s\n", rank, hf_last_error(NULL)); goto done; }
    hf_hostio_config hc;
    memset(&hc, 0, sizeof(hc));
    hc.struct_size = sizeof(hc);
    hc.frame_output_mode = HF_MODE_BLENDED_FRAME;
    hc.scene_change_threshold = threshold;
    hc.source_frame_time = src_t; hc.target_frame_time = tgt_t;
    if (hf_hostio_create(ctx, &hc, &hio)) { fprintf(stderr, "rank %d: %s\n", rank, hf_hostio_last_error(NULL)); goto done; }
    rc = hf_hostio_run(hio, &ch, n_out, t, fill, sink, &io, NULL) ? 1 : 0;
    if (rc) fprintf(stderr, "rank %d: %s\n", rank, hf_hostio_last_error(hio));
    uint64_t bi = 0, bo = 0;
    hf_hostio_get_traffic(hio, &bi, &bo);
    fprintf(stderr, "rank %d/%d device %d: source periods %lld..%lld (+%lld warm-up frames) -> %lld output frames from #%lld (%d copies), %.1f MB up, %.1f MB down\n",
            rank, world, hf_get_device(ctx), (long long)ch.first_period, (long long)(ch.first_period + ch.n_periods - 1),
            (long long)(ch.first_period - ch.first_frame), (long long)ch.n_outputs, (long long)ch.first_output, io.copies, bi / 1e6, bo / 1e6);
done:
    if (hio) hf_hostio_destroy(hio);
    if (ctx) hf_destroy(ctx);
    if (io.fd_in >= 0) close(io.fd_in);
    if (io.fd_out >= 0) close(io.fd_out);
    free(n_out); free(t);
    return rc;
}

int main(int argc, char** argv) {
    if (argc == 14 && strcmp(argv[1], "--worker") == 0)      /* re-executed by the parent below: one rank */
        return worker(argv[2], argv[3], atoi(argv[4]), atoi(argv[5]), atoi(argv[6]), atoll(argv[7]), atoll(argv[8]), atoll(argv[9]), atoi(argv[10]),
                      atoi(argv[11]), atoi(argv[12]), atoi(argv[13]));
    if (argc < 8) { fprintf(stderr, "usage: %s in out WIDTH HEIGHT HDR TARGET_FPS GPUS [RADIUS] [SCENE_THRESHOLD]\n", argv[0]); return 2; }
    const char *in = argv[1], *out = argv[2];
    const int W = atoi(argv[3]), H = atoi(argv[4]), hdr = atoi(argv[5]), gpus = atoi(argv[7]);
    const double target_fps = atof(argv[6]);
    const int radius = argc > 8 ? atoi(argv[8]) : 16, threshold = argc > 9 ? atoi(argv[9]) : -1;
    const int64_t src_t = 417083, tgt_t = (int64_t)(1e7 / target_fps + 0.5);   /* 100-ns units (HopperRender.cpp:162-163) */
    struct stat st;
    if (stat(in, &st) != 0) { perror(in); return 1; }
    const size_t frame_bytes = (size_t)W * (size_t)H * 3 / 2 * (hdr ? 2 : 1);
    const int64_t n_frames = (int64_t)(st.st_size / (off_t)frame_bytes);
    hf_timeline_chunk all;                                   /* the whole clip as one chunk: the number of output frames */
    if (hf_shard_timeline(n_frames, 1, 0, src_t, tgt_t, 3, 12, &all, NULL, NULL, 0)) { fprintf(stderr, "%s\n", hf_hostio_last_error(NULL)); return 1; }
    const int fd = open(out, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0 || ftruncate(fd, (off_t)all.n_outputs * (off_t)frame_bytes) != 0) { perror(out); return 1; }
    close(fd);
    int failed = 0;
    for (int r = 0; r < gpus; r++) {                        /* one fresh process per GPU (fork + exec of this program), started before any GPU call */
        const pid_t pid = fork();
        if (pid == 0) {
            char a[10][32];
            snprintf(a[0], 32, "%d", W); snprintf(a[1], 32, "%d", H); snprintf(a[2], 32, "%d", hdr); snprintf(a[3], 32, "%lld", (long long)src_t);
            snprintf(a[4], 32, "%lld", (long long)tgt_t); snprintf(a[5], 32, "%lld", (long long)n_frames); snprintf(a[6], 32, "%d", gpus);
            snprintf(a[7], 32, "%d", r); snprintf(a[8], 32, "%d", radius); snprintf(a[9], 32, "%d", threshold);
            execl("/proc/self/exe", argv[0], "--worker", in, out, a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], (char*)NULL);
            perror("execl");
            _exit(127);
        }
        if (pid < 0) { perror("fork"); failed = 1; }
    }
    for (int r = 0; r < gpus; r++) {
        int status = 0;
        if (wait(&status) < 0 || !WIFEXITED(status) || WEXITSTATUS(status) != 0) failed = 1;
    }
    fprintf(stderr, "%lld source frames -> %lld output frames on %d GPU worker(s)%s\n", (long long)n_frames, (long long)all.n_outputs, gpus,
            failed ? " -- FAILED" : "");
    return failed;
}
