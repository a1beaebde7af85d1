// sched_format.h -- bit layout of one wave-step descriptor, shared by the host
// schedule builder (host_core.cpp) and the kernels (kernels.hip).
//
// Every wave of a tile executes ONE linear stream of wave-steps per schedule
// (64 lanes x one slot).  A descriptor says what the step is and how many
// workgroup barriers the wave has to pass before touching the solve vector
// for it (= number of phase boundaries since the wave's previous step):
//   bits 0-2   lt      row step: groups of 2^lt lanes accumulate one target row
//   bit  3     flush   row step: reduce the groups and apply them to their rows
//   bits 4-5   type    0 row step, 1 block-task step, 3 no-op
//   bits 6-9   s       block step: index of the step inside its task (columns s*BT .. s*BT+BT-1)
//   bit  10    first   block step: load the task's rows first
//   bit  11    last    block step: store the task's rows afterwards
//   bits 12-31 nbar    barriers before this step
#pragma once
#define MI_D_LT(d) ((d) & 7u)
#define MI_D_FLUSH 8u
#define MI_D_TYPE(d) (((d) >> 4) & 3u)
#define MI_D_TYPE_ROW 0u
#define MI_D_TYPE_BLOCK 1u
#define MI_D_NOOP 0x30u
#define MI_D_S(d) (((d) >> 6) & 15u)
#define MI_D_FIRST 0x400u
#define MI_D_LAST 0x800u
#define MI_D_NBAR(d) ((d) >> 12)
#define MI_D_NBAR_MAX 0xFFFFFu
