// Many games per host thread: user-level fibers for the self-play games (go.cpp) and the leaf pool's blocking predict (batcher.cpp).
//
// The reference runs one OS thread per game plus the search workers (src/libdg_mcts/self_play.rs:478-495): at 256 games in flight
// per GPU that is 256 threads whose every leaf evaluation is a futex sleep and a futex wake-up.  Here a game can be a FIBER: a
// 1 MiB stack of its own, switched in user space (six callee-saved registers and the stack pointer, no system call), a fixed
// scheduler thread per fiber.  A fiber that calls the pool's predict queues its boards, switches to its scheduler and is made
// ready again by the pool worker that completes the request; the game, search and board code above is unchanged and still reads
// as blocking code.  x86-64 only; elsewhere (and under the sanitizers, -DDGNN_NO_FIBERS) fibers_available() is false and the
// callers fall back to one OS thread per game.
#pragma once

#include <functional>

namespace dgnn {

struct Fiber;

bool fibers_available();

// Runs `count` fibers (body(index) each) on `threads` scheduler threads (fiber i on thread i % threads); returns when all are done.
void run_fibers(int count, int threads, const std::function<void(int)>& body);

// The fiber the calling code runs on, or nullptr on a plain thread.
Fiber* fiber_current();
// Switches to the scheduler; returns after fiber_make_ready(this fiber) was called (from any thread, before or after the yield).
void fiber_yield();
void fiber_make_ready(Fiber* fiber);

}  // namespace dgnn
