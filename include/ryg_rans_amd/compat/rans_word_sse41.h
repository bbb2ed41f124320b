/* forwarding header: the reference include name -> our re-provision */
#include "rans_word_compat.h"
