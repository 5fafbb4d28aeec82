#include <vector>
#include <thread>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <sys/mman.h>
struct V { uint32_t id; int32_t a; uint32_t q; int32_t pos; };
int main(int argc, char** argv) {
  int threads = atoi(argv[1]); int huge = atoi(argv[2]);
  size_t reads = 100000; std::vector<std::vector<V>*> rs(reads); std::vector<uint64_t> ptr(reads + 1, 0);
  for (size_t r = 0; r < reads; ++r) { rs[r] = new std::vector<V>(36); for (auto& v : *rs[r]) v = {0, 1, 30, (int)r}; ptr[r + 1] = ptr[r] + 36; }
  size_t nnz = ptr[reads];
  for (int rep = 0; rep < 5; ++rep) {
    auto t0 = std::chrono::steady_clock::now();
    int32_t* pos; uint8_t* al; uint32_t* q;
    auto alloc = [&](size_t b) { void* p; if (huge) { size_t r = (b + (2 << 20) - 1) / (2 << 20) * (2 << 20); posix_memalign(&p, 2 << 20, r); madvise(p, r, MADV_HUGEPAGE); } else p = malloc(b); return p; };
    pos = (int32_t*)alloc(nnz * 4); al = (uint8_t*)alloc(nnz); q = (uint32_t*)alloc(nnz * 4);
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t) pool.emplace_back([&, t] { size_t lo = reads * t / threads, hi = reads * (t + 1) / threads;
      for (size_t r = lo; r < hi; ++r) { const V* raw = rs[r]->data(); size_t at = ptr[r]; for (int i = 0; i < 36; ++i, ++at) { pos[at] = raw[i].pos; al[at] = raw[i].a; q[at] = raw[i].q; } } });
    for (auto& th : pool) th.join();
    auto t1 = std::chrono::steady_clock::now();
    printf("threads %d huge %d: %.1f ms\n", threads, huge, std::chrono::duration<double, std::milli>(t1 - t0).count());
    free(pos); free(al); free(q);
  }
}
