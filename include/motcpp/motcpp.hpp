// Umbrella header (reference: include/motcpp/motcpp.hpp:28-37) for the MI355X-native hot path.
#pragma once
#include "tracker.hpp"
#include "device_tracker.hpp"
#include "trackers/sort.hpp"
#include "trackers/bytetrack.hpp"
#include "trackers/ocsort.hpp"
#include "trackers/botsort.hpp"
#include "trackers/deepocsort.hpp"
#include "trackers/strongsort.hpp"
#include "trackers/ucmc.hpp"
#include "trackers/boosttrack.hpp"
#include "utils/matching.hpp"
#include "utils/iou.hpp"
