// STUB (see Reference.h)
#include "Reference.h"
