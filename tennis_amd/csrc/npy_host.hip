// Host side of `evaluate.py --save_feats` (reference evaluate.py:306-321): one float32 `.npy` file per frame at
// dataset.save_feature_path(idx), skipped when the file exists.  The reference writes them with np.save from the evaluation loop;
// at the encoder's rate that loop IS the job: 13 ms of Python per 256 files against 1.8 ms of GPU time for their features
// (19 k frames/s, measured round 4).  This is the writer behind the C ABI: a pool of threads that creates the directories and
// writes NumPy format 1.0 files byte for byte as np.save does, fed batch by batch while the GPU encodes the next one.
// Pure host code (no GPU needed): tests/test_cpu_oracle.py compares the files with np.save's.
#include <fcntl.h>
#include <stdio.h>      // renameat2, RENAME_NOREPLACE
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cerrno>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "common.h"

namespace {

// the header np.save writes for a C-contiguous float32 vector of `dim` elements (format 1.0: magic, version, uint16 length,
// a dict literal padded with spaces and closed by '\n' so that data starts at a multiple of 64 bytes)
std::string npy_header_f32(int dim) {
  std::string d = "{'descr': '<f4', 'fortran_order': False, 'shape': (" + std::to_string(dim) + ",), }";
  const size_t unpadded = 10 + d.size() + 1;
  const size_t pad = (64 - unpadded % 64) % 64;
  d.append(pad, ' ');
  d.push_back('\n');
  std::string h("\x93NUMPY\x01\x00", 8);
  h.push_back((char)(d.size() & 0xff));
  h.push_back((char)(d.size() >> 8));
  return h + d;
}

constexpr int kMaxQueuedJobs = 32;

struct Job {
  std::vector<float> rows;
  std::vector<std::string> paths;
  int dim = 0;
  bool skip_existing = true;
  int next = 0;                      // next row to hand out (under the writer's lock)
};

}  // namespace

struct tn_npy_writer {
  std::vector<std::thread> pool;
  std::mutex mu;
  std::condition_variable cv_work, cv_idle;
  std::deque<std::shared_ptr<Job>> queue;      // jobs with rows left to take
  int busy = 0;
  bool stop = false;
  std::set<std::string> dirs;                  // directories known to exist
  std::mutex dir_mu;
  std::atomic<long long> written{0}, skipped{0};
  std::mutex err_mu;
  std::string error;

  bool make_dirs(const std::string &path) {      // mkdir -p of the file's directory
    const size_t slash = path.rfind('/');
    if (slash == std::string::npos || slash == 0) return true;
    const std::string dir = path.substr(0, slash);
    {
      std::lock_guard<std::mutex> g(dir_mu);
      if (dirs.count(dir)) return true;
    }
    for (size_t p = 1; p <= dir.size(); ++p) {
      if (p != dir.size() && dir[p] != '/') continue;
      const std::string part = dir.substr(0, p);
      if (mkdir(part.c_str(), 0777) != 0 && errno != EEXIST) return false;
    }
    std::lock_guard<std::mutex> g(dir_mu);
    dirs.insert(dir);
    return true;
  }
  void fail(const std::string &what) {
    std::lock_guard<std::mutex> g(err_mu);
    if (error.empty()) error = what;
  }
  // A file appears under its name only when it is complete: the bytes go to "<path>.tmp.<pid>.<thread>" and the
  // file is then renamed (atomic within a directory).  A short write (ENOSPC, EIO) or a killed run leaves no truncated .npy
  // that the next --save_feats run would count as "exists" and the temporal head's training would read.
  void write_row(Job &j, int i, const std::string &header) {
    const std::string &path = j.paths[i];
    if (!make_dirs(path)) { fail("cannot create the directory of " + path + ": " + strerror(errno)); return; }
    if (j.skip_existing && access(path.c_str(), F_OK) == 0) { ++skipped; return; }          // evaluate.py:312: `if not os.path.exists(feat_path)`
    const std::string tmp = path + ".tmp." + std::to_string((long)getpid()) + "." + std::to_string((unsigned long)std::hash<std::thread::id>{}(std::this_thread::get_id()));
    const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
    if (fd < 0) {
      fail("cannot open " + tmp + ": " + strerror(errno));
      return;
    }
    std::string buf = header;
    buf.append((const char *)(j.rows.data() + (size_t)i * j.dim), sizeof(float) * j.dim);
    size_t off = 0;
    while (off < buf.size()) {
      const ssize_t n = write(fd, buf.data() + off, buf.size() - off);
      if (n < 0) {
        if (errno == EINTR) continue;
        fail("write to " + tmp + " failed: " + strerror(errno));
        break;
      }
      off += (size_t)n;
    }
    const bool closed = close(fd) == 0;
    if (off != buf.size() || !closed) {
      if (off == buf.size()) fail("close of " + tmp + " failed: " + strerror(errno));
      unlink(tmp.c_str());
      return;
    }
    if (j.skip_existing) {
      // another writer may have completed the same file meanwhile: link() fails with EEXIST instead of replacing it
      if (link(tmp.c_str(), path.c_str()) != 0) {
        int e = errno;
        // file systems without hard links (FAT / exFAT, several FUSE, object-store and SMB mounts: EPERM / ENOTSUP / EMLINK / EXDEV):
        // an atomic no-replace rename where the kernel has one, else "is it there?" + rename (every row failed there)
        if (e == EPERM || e == ENOTSUP || e == EOPNOTSUPP || e == EMLINK || e == EXDEV || e == ENOSYS) {
          if (renameat2(AT_FDCWD, tmp.c_str(), AT_FDCWD, path.c_str(), RENAME_NOREPLACE) == 0) { ++written; return; }
          e = errno;
          if (e != EEXIST) {
            if (access(path.c_str(), F_OK) == 0) e = EEXIST;
            else if (rename(tmp.c_str(), path.c_str()) == 0) { ++written; return; }
            else e = errno;
          }
        }
        unlink(tmp.c_str());
        if (e == EEXIST) { ++skipped; return; }
        fail("cannot publish " + path + ": " + strerror(e));
        return;
      }
      unlink(tmp.c_str());
    } else if (rename(tmp.c_str(), path.c_str()) != 0) {
      fail("cannot rename " + tmp + " to " + path + ": " + strerror(errno));
      unlink(tmp.c_str());
      return;
    }
    ++written;
  }
  void worker() {
    std::string header;
    int header_dim = -1;
    for (;;) {
      std::shared_ptr<Job> job;
      int i;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv_work.wait(lk, [&] { return stop || !queue.empty(); });
        if (queue.empty()) return;               // stop and nothing left
        job = queue.front();
        i = job->next++;                          // (claimed under the lock)
        if (i + 1 >= (int)job->paths.size()) queue.pop_front();
        ++busy;
      }
      if (header_dim != job->dim) { header = npy_header_f32(job->dim); header_dim = job->dim; }
      write_row(*job, i, header);
      {
        std::lock_guard<std::mutex> lk(mu);
        --busy;
      }
      cv_idle.notify_all();
    }
  }
};

extern "C" int tn_npy_writer_create(int threads, tn_npy_writer **out) {
  TN_REQUIRE(out, "tn_npy_writer_create: null argument");
  TN_REQUIRE(threads >= 1 && threads <= 256, "tn_npy_writer_create: threads must be in 1..256");
  tn_npy_writer *w = new tn_npy_writer();
  for (int t = 0; t < threads; ++t) w->pool.emplace_back([w] { w->worker(); });
  *out = w;
  return TN_OK;
}

// rows (n, dim) float32 in HOST memory, paths[n]: queued and written by the pool; the rows are copied before the call returns
extern "C" int tn_npy_writer_submit(tn_npy_writer *w, const float *rows, int n, int dim, const char *const *paths, int skip_existing) {
  TN_REQUIRE(w && rows && paths, "tn_npy_writer_submit: null argument");
  TN_REQUIRE(n > 0 && dim > 0, "tn_npy_writer_submit: bad shape");
  auto job = std::make_shared<Job>();
  job->rows.assign(rows, rows + (size_t)n * dim);
  job->paths.reserve(n);
  for (int i = 0; i < n; ++i) {
    TN_REQUIRE(paths[i] && paths[i][0], "tn_npy_writer_submit: empty path");
    job->paths.emplace_back(paths[i]);
  }
  job->dim = dim;
  job->skip_existing = skip_existing != 0;
  {
    // back-pressure: a disk slower than the encoder must not turn the queue into a copy of the feature set
    std::unique_lock<std::mutex> lk(w->mu);
    w->cv_idle.wait(lk, [&] { return (int)w->queue.size() < kMaxQueuedJobs; });
    w->queue.push_back(job);
  }
  w->cv_work.notify_all();
  return TN_OK;
}

// waits until everything submitted so far is on disk; totals since creation; the first error of any file fails the call
extern "C" int tn_npy_writer_drain(tn_npy_writer *w, int64_t *written, int64_t *skipped) {
  TN_REQUIRE(w, "tn_npy_writer_drain: null handle");
  {
    std::unique_lock<std::mutex> lk(w->mu);
    w->cv_idle.wait(lk, [&] { return w->queue.empty() && w->busy == 0; });
  }
  if (written) *written = w->written.load();
  if (skipped) *skipped = w->skipped.load();
  std::lock_guard<std::mutex> g(w->err_mu);
  if (!w->error.empty()) {
    tn_set_error("tn_npy_writer: " + w->error);
    w->error.clear();
    return TN_ERR_INVALID;
  }
  return TN_OK;
}

extern "C" int tn_npy_writer_destroy(tn_npy_writer *w) {
  if (!w) return TN_OK;
  {
    std::unique_lock<std::mutex> lk(w->mu);
    w->cv_idle.wait(lk, [&] { return w->queue.empty() && w->busy == 0; });
    w->stop = true;
  }
  w->cv_work.notify_all();
  for (auto &t : w->pool) t.join();
  delete w;
  return TN_OK;
}
