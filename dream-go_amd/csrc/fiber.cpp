// User-level fibers (fiber.h).  Host code only.
#include "fiber.h"
#include "tsan_mutex.h"

#include <sys/mman.h>

#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <vector>

#if defined(__x86_64__) && !defined(DGNN_NO_FIBERS)
#define DGNN_FIBERS 1
#else
#define DGNN_FIBERS 0
#endif

// The sanitizers follow a hand-written stack switch when they are told about it (round 4: what ships is what is sanitized):
// ASan needs the bounds of the stack that is switched TO and a slot for the fake stack of the one that is left
// (__sanitizer_start/finish_switch_fiber), TSan a context per fiber (__tsan_create_fiber / __tsan_switch_to_fiber).
#if defined(__SANITIZE_ADDRESS__)
#include <sanitizer/common_interface_defs.h>
#define DGNN_ASAN_FIBERS 1
#else
#define DGNN_ASAN_FIBERS 0
#endif
#if defined(__SANITIZE_THREAD__)
#include <sanitizer/tsan_interface.h>
#define DGNN_TSAN_FIBERS 1
#else
#define DGNN_TSAN_FIBERS 0
#endif

#if DGNN_FIBERS && !defined(__HIP_DEVICE_COMPILE__)
// void dgnn_fiber_switch(void** save_sp, void* load_sp): System V x86-64, callee-saved registers only (the switch is a call:
// everything else is dead across it); the floating-point control words are the same in every fiber and are not touched.
asm(R"(
.text
.globl dgnn_fiber_switch
.hidden dgnn_fiber_switch
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
    void* asan_fake_stack = nullptr;           // sanitizer builds only: ASan's fake stack while switched out, TSan's fiber context
    void* tsan_fiber = nullptr;
};

namespace {

struct Scheduler {
    std::mutex mutex;
    dgnn::MutexLife mutex_life{&mutex};
    std::condition_variable cv;
    std::vector<Fiber*> ready;                 // pushed by any thread
    bool sleeping = false;
    void* sp = nullptr;                        // the scheduler thread's own context while a fiber runs
    Fiber* current = nullptr;
    int live = 0;
    void* asan_fake_stack = nullptr;           // sanitizer builds only: the scheduler thread's own contexts
    const void* asan_stack_bottom = nullptr;
    size_t asan_stack_size = 0;
    void* tsan_fiber = nullptr;
};

thread_local Scheduler* tls_scheduler = nullptr;

#if DGNN_FIBERS
// fiber -> its scheduler
void switch_to_scheduler(Scheduler* s, Fiber* f, bool for_good)
{
#if DGNN_ASAN_FIBERS
    __sanitizer_start_switch_fiber(for_good ? nullptr : &f->asan_fake_stack, s->asan_stack_bottom, s->asan_stack_size);
#endif
#if DGNN_TSAN_FIBERS
    __tsan_switch_to_fiber(s->tsan_fiber, 0);
#endif
    (void)for_good;
    dgnn_fiber_switch(&f->sp, s->sp);
#if DGNN_ASAN_FIBERS
    __sanitizer_finish_switch_fiber(f->asan_fake_stack, &s->asan_stack_bottom, &s->asan_stack_size);     // resumed by the scheduler
#endif
}

// scheduler -> fiber
void switch_to_fiber(Scheduler* s, Fiber* f)
{
#if DGNN_ASAN_FIBERS
    __sanitizer_start_switch_fiber(&s->asan_fake_stack, static_cast<char*>(f->stack) + kGuardBytes, kStackBytes);
#endif
#if DGNN_TSAN_FIBERS
    __tsan_switch_to_fiber(f->tsan_fiber, 0);
#endif
    dgnn_fiber_switch(&s->sp, f->sp);
#if DGNN_ASAN_FIBERS
    __sanitizer_finish_switch_fiber(s->asan_fake_stack, nullptr, nullptr);
#endif
}

void fiber_entry()
{
    Scheduler* s = tls_scheduler;
    Fiber* f = s->current;
#if DGNN_ASAN_FIBERS
    __sanitizer_finish_switch_fiber(nullptr, &s->asan_stack_bottom, &s->asan_stack_size);     // first time on this stack: learn the scheduler's
#endif
    (*f->body)(f->index);
    f->finished = true;
    switch_to_scheduler(s, f, true);           // never resumed
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
#if DGNN_TSAN_FIBERS
    f->tsan_fiber = __tsan_create_fiber(0);
#endif
    return true;
}

void scheduler_loop(Scheduler* s)
{
    tls_scheduler = s;
#if DGNN_TSAN_FIBERS
    s->tsan_fiber = __tsan_get_current_fiber();
#endif
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
            switch_to_fiber(s, f);
            s->current = nullptr;
            if (f->finished) {
#if DGNN_TSAN_FIBERS
                __tsan_destroy_fiber(f->tsan_fiber);
#endif
                munmap(f->stack, kStackBytes + kGuardBytes);
                f->stack = nullptr;
                --s->live;
            }
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
    // The notify happens UNDER the mutex: the Scheduler lives on run_fibers' stack, and once the lock is released the scheduler
    // thread may run the last fibers to completion (woken by another waker or spuriously) and destroy it -- a notify after the
    // unlock could then signal a dead condition variable.
    Scheduler* s = f->scheduler;
    std::lock_guard<std::mutex> lock(s->mutex);
    s->ready.push_back(f);
    if (s->sleeping) s->cv.notify_one();
}

void fiber_yield()
{
#if DGNN_FIBERS
    Scheduler* s = tls_scheduler;
    Fiber* f = s->current;
    switch_to_scheduler(s, f, false);
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
