// Host-side sweep-file reader pool.
//
// The reference feeds its sweep through 8 DataLoader worker PROCESSES, each of which np.fromfile()s the
// key-frame .bin and nine sweep .bin files of a sample, slices / filters / transforms them with numpy and
// pickles the result back to the parent (det3d/datasets/loader/build_loader.py:23-59,
// det3d/datasets/pipelines/loading.py:17-63,98-126).  Here the files' bytes go straight from the page
// cache into a PINNED staging buffer owned by the caller -- one pread() per file, files of a whole batch in
// parallel on a small thread pool -- and everything else (column cut, remove_close, float64 transform,
// time column, compaction) happens on the device in al3d_merge_sweeps_batch_f32.  The pool is
// asynchronous: submit() returns a job id at once, wait() blocks until that batch has landed, so batch
// i+1 is read while batch i is uploaded and convolved.
//
// Plain C ABI (include/al3d.h); no torch, no HIP calls.  Errors raised on a worker thread are kept in the
// job and handed to the waiting thread's al3d_last_error().
#include <errno.h>
#include <fcntl.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include <deque>
#include <map>
#include <string>
#include <vector>

#include "al3d_common.h"

namespace {

struct FileTask {
    std::string path;
    int64_t rows;          // whole 20-byte rows to read
    unsigned char* dst;
};

struct Job {
    std::vector<FileTask> tasks;
    size_t next = 0;       // next task to hand out
    size_t done = 0;       // tasks finished
    int status = 0;
    std::string error;
};

}  // namespace

struct al3d_reader {
    pthread_mutex_t mu;
    pthread_cond_t work_cv, done_cv;
    std::vector<pthread_t> threads;
    std::deque<int> queue;            // job ids with tasks left to hand out
    std::map<int, Job> jobs;
    int next_id = 0;
    bool stop = false;
};

static int read_rows(const FileTask& t, std::string* err)
{
    if (t.rows == 0) return 0;
    const int fd = open(t.path.c_str(), O_RDONLY);
    if (fd < 0) {
        *err = "open " + t.path + ": " + strerror(errno);
        return -1;
    }
    int64_t want = t.rows * 20, got = 0;
    while (got < want) {
        const ssize_t r = pread(fd, t.dst + got, (size_t)(want - got), (off_t)got);
        if (r < 0 && errno == EINTR) continue;
        if (r <= 0) {
            *err = "read " + t.path + ": " + (r < 0 ? strerror(errno) : "file shrank while it was read");
            close(fd);
            return -1;
        }
        got += r;
    }
    close(fd);
    return 0;
}

static void* reader_main(void* arg)
{
    al3d_reader* rd = (al3d_reader*)arg;
    pthread_mutex_lock(&rd->mu);
    for (;;) {
        while (!rd->stop && rd->queue.empty()) pthread_cond_wait(&rd->work_cv, &rd->mu);
        if (rd->stop) break;
        const int id = rd->queue.front();
        Job& job = rd->jobs[id];
        const size_t ti = job.next++;
        if (job.next >= job.tasks.size()) rd->queue.pop_front();
        const FileTask task = job.tasks[ti];
        pthread_mutex_unlock(&rd->mu);
        std::string err;
        const int rc = read_rows(task, &err);
        pthread_mutex_lock(&rd->mu);
        Job& j2 = rd->jobs[id];
        if (rc != 0 && j2.status == 0) {
            j2.status = rc;
            j2.error = err;
        }
        if (++j2.done == j2.tasks.size()) pthread_cond_broadcast(&rd->done_cv);
    }
    pthread_mutex_unlock(&rd->mu);
    return nullptr;
}

extern "C" int al3d_reader_create(int n_threads, al3d_reader** out)
{
    AL3D_REQUIRE(out && n_threads >= 1 && n_threads <= 256, "al3d_reader_create: 1..256 threads, non-null out");
    al3d_reader* rd = new al3d_reader();
    pthread_mutex_init(&rd->mu, nullptr);
    pthread_cond_init(&rd->work_cv, nullptr);
    pthread_cond_init(&rd->done_cv, nullptr);
    rd->threads.resize((size_t)n_threads);
    for (int i = 0; i < n_threads; ++i) {
        if (pthread_create(&rd->threads[(size_t)i], nullptr, reader_main, rd) != 0) {
            rd->threads.resize((size_t)i);
            al3d_reader_destroy(rd);
            return al3d_fail(AL3D_ELAUNCH, "al3d_reader_create: pthread_create failed");
        }
    }
    *out = rd;
    return AL3D_OK;
}

extern "C" void al3d_reader_destroy(al3d_reader* rd)
{
    if (!rd) return;
    pthread_mutex_lock(&rd->mu);
    rd->stop = true;
    pthread_cond_broadcast(&rd->work_cv);
    pthread_mutex_unlock(&rd->mu);
    for (pthread_t& t : rd->threads) pthread_join(t, nullptr);
    pthread_mutex_destroy(&rd->mu);
    pthread_cond_destroy(&rd->work_cv);
    pthread_cond_destroy(&rd->done_cv);
    delete rd;
}

// rows_out[i] = whole float32 x,y,z,intensity,ring rows of file i (the reference's read_file drops a
// trailing partial row: loading.py:17-24); returns the total or a negative status.
extern "C" int64_t al3d_reader_plan(const char* const* paths, int n_files, int64_t* rows_out)
{
    if (!paths || !rows_out || n_files < 0) return al3d_fail(AL3D_EINVAL, "al3d_reader_plan: bad arguments");
    int64_t total = 0;
    for (int i = 0; i < n_files; ++i) {
        struct stat st;
        if (!paths[i] || stat(paths[i], &st) != 0)
            return al3d_fail(AL3D_EINVAL, "al3d_reader_plan: %s: %s", paths[i] ? paths[i] : "(null)", strerror(errno));
        rows_out[i] = (int64_t)(st.st_size / 20);
        total += rows_out[i];
    }
    return total;
}

// Asynchronously read rows[i] rows of paths[i] to dst + 20 * row_off[i].  Returns a job id >= 0.
extern "C" int al3d_reader_submit(al3d_reader* rd, const char* const* paths, int n_files, const int64_t* row_off,
                                  const int64_t* rows, void* dst, int64_t dst_bytes)
{
    AL3D_REQUIRE(rd && paths && row_off && rows && n_files >= 0 && (dst || n_files == 0),
                 "al3d_reader_submit: null pointer");
    Job job;
    job.tasks.reserve((size_t)n_files);
    for (int i = 0; i < n_files; ++i) {
        AL3D_REQUIRE(paths[i] && rows[i] >= 0 && row_off[i] >= 0 && (row_off[i] + rows[i]) * 20 <= dst_bytes,
                     "al3d_reader_submit: file %d does not fit the destination buffer", i);
        job.tasks.push_back(FileTask{paths[i], rows[i], (unsigned char*)dst + row_off[i] * 20});
    }
    pthread_mutex_lock(&rd->mu);
    const int id = rd->next_id++;
    const bool empty = job.tasks.empty();
    rd->jobs[id] = std::move(job);
    if (!empty) {
        rd->queue.push_back(id);
        pthread_cond_broadcast(&rd->work_cv);
    }
    pthread_mutex_unlock(&rd->mu);
    return id;
}

// Block until every file of the job has been read; the job is forgotten afterwards.
extern "C" int al3d_reader_wait(al3d_reader* rd, int job_id)
{
    AL3D_REQUIRE(rd, "al3d_reader_wait: null reader");
    pthread_mutex_lock(&rd->mu);
    auto it = rd->jobs.find(job_id);
    if (it == rd->jobs.end()) {
        pthread_mutex_unlock(&rd->mu);
        return al3d_fail(AL3D_EINVAL, "al3d_reader_wait: unknown job %d", job_id);
    }
    while (it->second.done < it->second.tasks.size()) pthread_cond_wait(&rd->done_cv, &rd->mu);
    const int status = it->second.status;
    const std::string err = it->second.error;
    rd->jobs.erase(it);
    pthread_mutex_unlock(&rd->mu);
    if (status != 0) return al3d_fail(AL3D_EINVAL, "al3d_reader: %s", err.c_str());
    return AL3D_OK;
}
