// What ThreadSanitizer has to be told (tools/sanitize_host.sh) so that its run is silent BY CONSTRUCTION; nothing here changes a shipped build.
#pragma once
#include <chrono>
#include <condition_variable>
#include <mutex>
#if defined(__SANITIZE_THREAD__)
#include <sanitizer/tsan_interface.h>
#endif

namespace dgnn {

// std::mutex has a constexpr constructor: ThreadSanitizer never sees a mutex of a NEW object come to life, and when the allocator
// hands a pool / request / scheduler the address of an earlier one it keeps treating the mutex there as the old, destroyed one --
// "data race" reports between two threads that both hold it.  A MutexLife member declared right behind a mutex announces its
// creation and destruction to TSan (the sanitizer build only; nothing in the shipped code).
#if defined(__SANITIZE_THREAD__)
struct MutexLife {
    void* m;
    explicit MutexLife(void* p) : m(p) { __tsan_mutex_create(m, 0); }
    ~MutexLife() { __tsan_mutex_destroy(m, 0); }
    MutexLife(const MutexLife&) = delete;
    MutexLife& operator=(const MutexLife&) = delete;
};
#else
struct MutexLife { explicit MutexLife(void*) {} };
#endif

// A timed wait on a condition variable.  libstdc++ waits on the steady clock with pthread_cond_clockwait, which this toolchain's TSan
// runtime does not intercept: it misses the unlock / relock inside the wait and reports "races" between two threads that both hold
// the mutex.  Under TSan the same wait goes through pthread_cond_timedwait (system clock), which it does intercept.
template <typename Pred>
inline bool wait_for_us(std::condition_variable& cv, std::unique_lock<std::mutex>& lock, long long us, Pred pred)
{
#if defined(__SANITIZE_THREAD__)
    return cv.wait_until(lock, std::chrono::system_clock::now() + std::chrono::microseconds(us), pred);
#else
    return cv.wait_for(lock, std::chrono::microseconds(us), pred);
#endif
}

}  // namespace dgnn
