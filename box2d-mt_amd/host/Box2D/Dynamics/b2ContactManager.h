// Drop-in header: what user code reads of the contact manager (reference: Box2D/Dynamics/b2ContactManager.h:66-180; the
// Testbed's Tiles scene asks m_broadPhase for the tree height). Contacts are managed on the device: this is a view.
#ifndef B2_CONTACT_MANAGER_H
#define B2_CONTACT_MANAGER_H

#include "Box2D/Collision/b2BroadPhase.h"

class b2Contact;
class b2ContactFilter;
class b2ContactListener;

class b2ContactManager
{
public:
	explicit b2ContactManager(b2World* world) : m_broadPhase(world), m_contactFilter(nullptr), m_contactListener(nullptr) {}
	b2BroadPhase m_broadPhase;
	b2ContactFilter* m_contactFilter;
	b2ContactListener* m_contactListener;
};

#endif
