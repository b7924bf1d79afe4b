// Enumerations of include/rodeo_kalman.h needed by device code (repeated here so that hiprtc translation units do not
// need the C header; the values are checked against the header in api.hip).
#pragma once
#ifndef RK_INTERROGATE_RODEO
#define RK_INTERROGATE_RODEO      0
#define RK_INTERROGATE_SCHOBER    1
#define RK_INTERROGATE_KRAMER     2
#define RK_INTERROGATE_CHKREBTII  3
#endif
