// ska_main.cpp -- the `ska` executable (build | align | distance | nk | ...) of the MI355X engine.
#include "../../include/skx_host.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <link.h>
#include <unistd.h>
// skh_main has released what it held (array, context) and written its outputs; what is left at this point is the HIP runtime's own
// exit-time teardown (queues, code objects, its memory pools: ~0.1 s), which a process that is about to disappear does not need: streams
// flushed, then _exit.  The fast path is NOT taken when something in the process depends on atexit handlers or static destructors: a
// profiler / tracer / sanitizer / coverage runtime mapped into the process (rocprofv3, roctracer, ASan, gcov ...) or announced in the
// environment, or SKX_KEEP_TEARDOWN=1 (always the normal return); SKX_FAST_EXIT=1 forces it.
static int tool_object(struct dl_phdr_info *info, size_t, void *found)
{
    static const char *const marks[] = {"rocprofiler", "roctracer", "roctx", "libasan", "libtsan", "libubsan", "liblsan", "clang_rt", "libgcov", "libprofiler", "valgrind", "vgpreload"};
    const char *n = info->dlpi_name ? info->dlpi_name : "";
    for (const char *m : marks) if (strstr(n, m)) { *(int *)found = 1; return 1; }
    return 0;
}
static bool fast_exit_allowed()
{
    if (getenv("SKX_KEEP_TEARDOWN")) return false;
    if (getenv("SKX_FAST_EXIT")) return true;
    for (const char *v : {"HSA_TOOLS_LIB", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD", "ROCP_TOOL_LIB", "LD_PRELOAD", "ASAN_OPTIONS", "LSAN_OPTIONS",
                          "GCOV_PREFIX", "LLVM_PROFILE_FILE"})
        if (const char *s = getenv(v)) if (*s) return false;
    int found = 0;
    dl_iterate_phdr(tool_object, &found);
    return !found;
}
int main(int argc, char **argv)
{
    const int rc = skh_main(argc, argv);
    if (!fast_exit_allowed()) return rc;
    fflush(nullptr);
    _exit(rc);
}
