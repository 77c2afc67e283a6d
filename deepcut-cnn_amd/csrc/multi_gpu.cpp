// multi_gpu.cpp — placeholder filled below
#include "net_internal.h"
