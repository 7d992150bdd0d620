#ifndef LVREF_TF2MSG_STUB
#define LVREF_TF2MSG_STUB
namespace tf2_msgs { struct TFMessage {}; }
#endif
