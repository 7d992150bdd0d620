#ifndef LVREF_STD_BOOL_STUB
#define LVREF_STD_BOOL_STUB
namespace std_msgs { struct Bool { bool data = false; }; }
#endif
