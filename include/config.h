// include/config.h -- the compile-time constants the reference's filter and calculator share
// (reference HopperRender/config.h:4-29).  The filter never includes config.h itself: it reaches it through
// opticalFlowCalc.h (opticalFlowCalc.h:8) and uses MIN_/MAX_SEARCH_RADIUS, UPPER_/LOWER_PERF_BUFFER, the DEFAULT_*
// values and MAX_CALC_RES (HopperRender.cpp:180-183,1445-1454,1529-1569), so the drop-in header chain has to provide
// the same macro names with the same values.  When this directory precedes the filter's own on the include path this
// file IS that config.h; when the filter's own copy is found first, the `#ifndef` guards below keep its values.
// libhopperflow.so is built against these values as well (hf_context.hip ..., hf_filter.cpp).
#pragma once

// Quality
#ifndef MAX_CALC_RES
#define MAX_CALC_RES 270          // largest height of the flow grid (config.h:4)
#endif
#ifndef NUM_ITERATIONS
#define NUM_ITERATIONS 0          // refinement levels; 0 = as many as the grid allows (config.h:6)
#endif
#ifndef MIN_SEARCH_RADIUS
#define MIN_SEARCH_RADIUS 5       // candidate count the governor never goes below (config.h:8)
#endif
#ifndef MAX_SEARCH_RADIUS
#define MAX_SEARCH_RADIUS 16      // ... and never exceeds (config.h:9)
#endif

// Performance governor (HopperRender.cpp:1438-1463)
#ifndef AUTO_SEARCH_RADIUS_ADJUST
#define AUTO_SEARCH_RADIUS_ADJUST 1
#endif
#ifndef UPPER_PERF_BUFFER
#define UPPER_PERF_BUFFER 1.4     // calc_time * 1.4 > source frame time -> radius - 1 (config.h:14)
#endif
#ifndef LOWER_PERF_BUFFER
#define LOWER_PERF_BUFFER 1.6     // calc_time * 1.6 < source frame time -> radius + 1 (config.h:15)
#endif
#ifndef CALC_TIME_INTERVAL
#define CALC_TIME_INTERVAL 240    // flow calculations per average/peak window (config.h:17)
#endif

// Debugging switches of the filter shell (unused by the calculator; kept so that filter code compiles)
#ifndef INC_APP_IND
#define INC_APP_IND 1
#endif
#ifndef SAVE_STATS
#define SAVE_STATS 0
#endif

// Defaults of the settings the filter loads from the registry (config.h:23-28)
#ifndef DEFAULT_DELTA_SCALAR
#define DEFAULT_DELTA_SCALAR 8
#endif
#ifndef DEFAULT_NEIGHBOR_SCALAR
#define DEFAULT_NEIGHBOR_SCALAR 6
#endif
#ifndef DEFAULT_BLACK_LEVEL
#define DEFAULT_BLACK_LEVEL 0
#endif
#ifndef DEFAULT_WHITE_LEVEL
#define DEFAULT_WHITE_LEVEL 255
#endif
#ifndef DEFAULT_SCENE_CHANGE_THRESHOLD
#define DEFAULT_SCENE_CHANGE_THRESHOLD 200
#endif
#ifndef DEFAULT_BUFFER_FRAMES
#define DEFAULT_BUFFER_FRAMES 0
#endif
