// tests/mock_rccl/stall.cpp -- TEST INFRASTRUCTURE: hold a HIP stream for a number of milliseconds (a host function
// queued with hipLaunchHostFunc sleeps; everything queued behind it waits) -- the stand-in, on REAL RCCL, for a
// collective whose peer never arrives: tests/test_gpu_rccl.py checks that a communicator with a time limit gives up.
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <thread>

static void stall_fn(void* arg) {
    std::this_thread::sleep_for(std::chrono::milliseconds((long)(intptr_t)arg));
}

extern "C" int stall_stream(void* stream, long ms) {
    return (int)hipLaunchHostFunc((hipStream_t)stream, stall_fn, (void*)(intptr_t)ms);
}
