/* Lets `#include "lz4frame.h"` / `C.include "<lz4frame.h>"` (Conduit.hsc:120-124, CTypes.hsc) resolve to the
 * MI355X library's declarations: add this directory to `include-dirs` instead of lz4/lib. */
#include "../lz4f_mi355x.h"
