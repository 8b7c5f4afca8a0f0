// sk_abi.h -- the exception barrier of the C ABI.
//
// Every extern "C" entry point of the library is a function-try-block: nothing that is thrown inside (std::bad_alloc /
// std::length_error from the containers the host side allocates with, or anything else) may unwind into a caller that is
// C, ctypes or Rust -- there it would be std::terminate, i.e. the death of the whole worker pool instead of an error on
// one call (the reference's contract: an error ends that stream only, soundkit-decoder/src/lib.rs:3131-3134).
// A caught exception becomes a status -- SK_ERR_OOM for std::bad_alloc, SK_ERR_INTERNAL for everything else -- and its
// text is kept per thread for sk_last_exception().  Header-only (inline variables) so that the test harnesses that
// compile single product sources get the same behaviour.
#pragma once
#include <atomic>
#include <cstdio>
#include <exception>
#include <new>
#include <stdexcept>

#include "../../include/soundkit_amd.h"

namespace sk {

inline std::atomic<int> g_abi_throw_after{-1};  // test hook (sk_debug_throw_after): >= 0 counts entries down, then one throws
inline std::atomic<int> g_abi_throw_kind{0};    // 0 std::bad_alloc, 1 std::length_error-like, 2 a non-std exception
inline thread_local char g_abi_text[256] = "";

struct AbiDebugThrow {};  // what kind 2 throws

[[gnu::noinline]] inline void abi_debug_point() {
    int left = g_abi_throw_after.load(std::memory_order_relaxed);
    while (left >= 0) {
        if (g_abi_throw_after.compare_exchange_weak(left, left - 1, std::memory_order_relaxed)) {
            if (left != 0) return;
            const int kind = g_abi_throw_kind.load(std::memory_order_relaxed);
            if (kind == 0) throw std::bad_alloc();
            if (kind == 1) throw std::length_error("sk_debug_throw_after");
            throw AbiDebugThrow{};
        }
    }
}

inline void abi_enter() {
    if (g_abi_throw_after.load(std::memory_order_relaxed) >= 0) abi_debug_point();
}

// called from a catch (...) handler: classifies the exception in flight
[[gnu::noinline]] inline int abi_caught(const char *entry) noexcept {
    int status = SK_ERR_INTERNAL;
    try {
        throw;
    } catch (const std::bad_alloc &) {
        status = SK_ERR_OOM;
        std::snprintf(g_abi_text, sizeof g_abi_text, "%s: std::bad_alloc", entry);
    } catch (const std::exception &e) {
        std::snprintf(g_abi_text, sizeof g_abi_text, "%s: %s", entry, e.what());
    } catch (...) {
        std::snprintf(g_abi_text, sizeof g_abi_text, "%s: exception of unknown type", entry);
    }
    return status;
}

inline const char *abi_message() noexcept { return g_abi_text; }

}  // namespace sk
