/* forwarding header: the reference include name -> our re-provision */
#include "rans64_compat.h"
