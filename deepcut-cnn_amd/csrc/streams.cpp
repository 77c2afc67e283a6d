// streams.cpp — placeholder filled below
#include "net_internal.h"
