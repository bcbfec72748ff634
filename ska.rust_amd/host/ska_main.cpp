// ska_main.cpp -- the `ska` executable (build | align | distance | nk) of the MI355X engine.
#include "../../include/skx_host.h"
int main(int argc, char **argv) { return skh_main(argc, argv); }
