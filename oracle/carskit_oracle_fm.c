/* placeholder translation unit: FM sweep restatement lands here */
#include "carskit_oracle.h"
