// sched_format.h -- bit layout of one wave-step descriptor, shared by the host
// schedule builder (host_core.cpp) and the kernels (kernels.hip).
//
// Every wave of a tile executes ONE linear stream of wave-steps per schedule
// (64 lanes x one slot).  A descriptor says what the step is and how many
// workgroup barriers the wave has to pass before touching the solve vector
// for it (= number of phase boundaries since the wave's previous step):
//   bits 0-2   lt      groups of 2^lt lanes accumulate one target row
//   bit  3     flush   reduce the groups and apply them to their rows
//   bits 4-5   type    0 row step, 3 no-op
//   bit  6     store   flush writes the sum (row = sum) instead of subtracting it (row -= sum)
//   bit  7     pre     the gather of this step was already issued by the step two positions earlier
//   bit  8     ahead   issue the gather of the step two positions later (the two steps in between and after
//                      are row steps of the same level without barriers in front of them)
//   bits 12-31 nbar    barriers before this step
#pragma once
#define MI_D_LT(d) ((d) & 7u)
#define MI_D_FLUSH 8u
#define MI_D_TYPE(d) (((d) >> 4) & 3u)
#define MI_D_TYPE_ROW 0u
#define MI_D_NOOP 0x30u
#define MI_D_STORE 0x40u
#define MI_D_PRE 0x80u
#define MI_D_AHEAD 0x100u
#define MI_D_LOOKAHEAD 2
#define MI_D_NBAR(d) ((d) >> 12)
#define MI_D_NBAR_MAX 0xFFFFFu
// value-source codes of a slot (Schedule::src and the maps derived from it)
#define MI_SRC_ZERO (-1)
#define MI_SRC_ONE (-2)
