// ska_main.cpp -- the `ska` executable (build | align | distance | nk | ...) of the MI355X engine.
#include "../../include/skx_host.h"
#include <cstdio>
#include <cstdlib>
#include <unistd.h>
// skh_main has released what it held (array, context) and written its outputs; what is left at this point is the HIP runtime's own
// exit-time teardown (queues, code objects, its memory pools), which a process that is about to disappear does not need: streams flushed, then
// _exit (SKX_KEEP_TEARDOWN=1: return through the runtime's atexit handlers as before).
int main(int argc, char **argv)
{
    const int rc = skh_main(argc, argv);
    if (getenv("SKX_KEEP_TEARDOWN")) return rc;
    fflush(nullptr);
    _exit(rc);
}
