// User-level fibers (fiber.h).  Host code only.
#include "fiber.h"

#include <sys/mman.h>

#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <vector>

#if defined(__x86_64__) && !defined(DGNN_NO_FIBERS) && !defined(__SANITIZE_ADDRESS__) && !defined(__SANITIZE_THREAD__)
#define DGNN_FIBERS 1
#else
#define DGNN_FIBERS 0
#endif

#if DGNN_FIBERS && !defined(__HIP_DEVICE_COMPILE__)
// void dgnn_fiber_switch(void** save_sp, void* load_sp): System V x86-64, callee-saved registers only (the switch is a call:
// everything else is dead across it); the floating-point control words are the same in every fiber and are not touched.
asm(R"(
.text
.globl dgnn_fiber_switch
.type dgnn_fiber_switch,@function
dgnn_fiber_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size dgnn_fiber_switch,.-dgnn_fiber_switch
)");
#endif
extern "C" void dgnn_fiber_switch(void** save_sp, void* load_sp);

namespace dgnn {

namespace {
constexpr size_t kStackBytes = 1u << 20;       // the ladder reader recurses up to 200 deep with a 3 KB board per level
constexpr size_t kGuardBytes = 4096;
struct Scheduler;
}  // namespace

struct Fiber {
    void* sp = nullptr;                        // saved stack pointer while switched out
    void* stack = nullptr;                     // mmap'd: guard page + kStackBytes
    Scheduler* scheduler = nullptr;
    const std::function<void(int)>* body = nullptr;
    int index = 0;
    bool finished = false;
};

namespace {

struct Scheduler {
    std::mutex mutex;
    std::condition_variable cv;
    std::vector<Fiber*> ready;                 // pushed by any thread
    bool sleeping = false;
    void* sp = nullptr;                        // the scheduler thread's own context while a fiber runs
    Fiber* current = nullptr;
    int live = 0;
};

thread_local Scheduler* tls_scheduler = nullptr;

#if DGNN_FIBERS
void fiber_entry()
{
    Scheduler* s = tls_scheduler;
    Fiber* f = s->current;
    (*f->body)(f->index);
    f->finished = true;
    dgnn_fiber_switch(&f->sp, s->sp);          // never resumed
    std::abort();
}

bool prepare(Fiber* f)
{
    void* mem = mmap(nullptr, kStackBytes + kGuardBytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE | MAP_STACK, -1, 0);
    if (mem == MAP_FAILED) return false;
    mprotect(mem, kGuardBytes, PROT_NONE);
    f->stack = mem;
    // what dgnn_fiber_switch pops: six registers, then the return address; at fiber_entry's first instruction the stack pointer
    // must be 8 below a 16-byte boundary, as after a call
    uintptr_t top = (reinterpret_cast<uintptr_t>(mem) + kGuardBytes + kStackBytes) & ~(uintptr_t)15;
    uint64_t* slot = reinterpret_cast<uint64_t*>(top - 8);
    *slot = 0;                                               // (fiber_entry never returns)
    *--slot = reinterpret_cast<uint64_t>(&fiber_entry);
    for (int i = 0; i < 6; ++i) *--slot = 0;
    f->sp = slot;
    return true;
}

void scheduler_loop(Scheduler* s)
{
    tls_scheduler = s;
    std::vector<Fiber*> batch;
    while (s->live > 0) {
        {
            std::unique_lock<std::mutex> lock(s->mutex);
            while (s->ready.empty()) { s->sleeping = true; s->cv.wait(lock); }
            s->sleeping = false;
            batch.swap(s->ready);
        }
        for (Fiber* f : batch) {
            s->current = f;
            dgnn_fiber_switch(&s->sp, f->sp);
            s->current = nullptr;
            if (f->finished) { munmap(f->stack, kStackBytes + kGuardBytes); f->stack = nullptr; --s->live; }
        }
        batch.clear();
    }
    tls_scheduler = nullptr;
}
#endif

}  // namespace

bool fibers_available() { return DGNN_FIBERS != 0; }

Fiber* fiber_current() { return tls_scheduler ? tls_scheduler->current : nullptr; }

void fiber_make_ready(Fiber* f)
{
    Scheduler* s = f->scheduler;
    bool wake;
    {
        std::lock_guard<std::mutex> lock(s->mutex);
        s->ready.push_back(f);
        wake = s->sleeping;
    }
    if (wake) s->cv.notify_one();
}

void fiber_yield()
{
#if DGNN_FIBERS
    Scheduler* s = tls_scheduler;
    Fiber* f = s->current;
    dgnn_fiber_switch(&f->sp, s->sp);
#endif
}

void run_fibers(int count, int threads, const std::function<void(int)>& body)
{
#if DGNN_FIBERS
    if (count <= 0) return;
    if (threads <= 0) threads = 1;
    if (threads > count) threads = count;
    std::vector<Scheduler> schedulers((size_t)threads);
    std::vector<Fiber> fibers((size_t)count);
    for (int i = 0; i < count; ++i) {
        Fiber& f = fibers[(size_t)i];
        f.scheduler = &schedulers[(size_t)(i % threads)];
        f.body = &body;
        f.index = i;
        if (!prepare(&f)) { std::fprintf(stderr, "dgnn: cannot map a fiber stack\n"); std::abort(); }
        f.scheduler->ready.push_back(&f);
        ++f.scheduler->live;
    }
    std::vector<std::thread> workers;
    for (int t = 1; t < threads; ++t) workers.emplace_back(scheduler_loop, &schedulers[(size_t)t]);
    scheduler_loop(&schedulers[0]);
    for (std::thread& w : workers) w.join();
#else
    (void)count; (void)threads; (void)body;
#endif
}

}  // namespace dgnn
