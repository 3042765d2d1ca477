// Forwarding header of the lite harness interface (same include name as the SLAMBench header it stands in for).
#include "sb_lite_all.h"
