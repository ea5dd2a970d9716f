// Recording interface between the launch functions and the stage-program builder (stage.hip).
#pragma once
#include "common.hip.h"

#define LD_STAGE_CONV3 1
#define LD_STAGE_GN 2
#define LD_STAGE_ARG_BYTES 320

bool ld_stage_recording();                       // a recording is open on this host thread
// hand a launch over to the open recording instead of launching it (returns LD_OK; a launch that cannot be recorded
// marks the recording failed, and ld_stage_end reports why)
int ld_stage_record(int kind, int variant, const void* args, size_t bytes, int gx, int gy, int gz, size_t lds);
int ld_stage_unsupported(const char* what);
