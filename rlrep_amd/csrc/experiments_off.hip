// Product build (no RLREP_BUILD_EXPERIMENTS): the launchers of the opt-in engines that were measured and not adopted -- rowprog.hip (row-block
// programs and their cluster form), xchain.hip (per-XCD persistent chains) -- exist only as stubs.  The
// program builder never emits their stages in this build (engine_internal.h rl_rowprog_enabled / Builder::chain_enabled are false).
#include "engine.h"

extern "C" int rl_launch_rowprog(const RpLaunch*, int, hipStream_t) { return -100; }
extern "C" int rl_rowprog_init() { return 0; }
extern "C" int rl_launch_xchain(const XcLaunch*, hipStream_t) { return -100; }
