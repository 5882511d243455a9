// Command-list runtime: while a list is being recorded on the calling thread, every
// launch entry point of the library appends itself (arguments captured by value) instead
// of launching; disyolo_cmdlist_run replays a range of the list on a stream with one C
// call.  This is how the host drives a ~400-launch training step without paying the
// interpreter per launch, and what a hipGraph capture of the step wraps.
#pragma once
#include <functional>

bool dy_recording();
int dy_record(std::function<int(void*)> fn);

// place after argument validation, before the first launch
#define DY_RECORD_OR_RUN(...)                   \
  do {                                          \
    if (dy_recording()) return dy_record(__VA_ARGS__); \
  } while (0)
