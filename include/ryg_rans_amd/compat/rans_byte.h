/* forwarding header: the reference include name -> our re-provision */
#include "rans_byte_compat.h"
