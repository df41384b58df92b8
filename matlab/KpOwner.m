classdef KpOwner < handle
    %KpOwner: lifetime of ONE kp_mex handle (context reference, dictionary, MPC problem, trajectory object, ...).
    %   Ksysid / Kmpc are value classes: they are copied freely and MATLAB calls no destructor on them.  A handle-class
    %   member IS destroyed - when the last copy of the value object that carries it (and the last lift closure that
    %   captured it) is gone - so KsysidHip / KmpcHip keep their device handles inside KpOwner objects and nothing has to be
    %   released by hand.  `parent` keeps the owner of the object a handle points into (a dictionary points into its
    %   context) alive until this one is deleted, which fixes the order of destruction.
    % Transient: a handle is a pointer of THIS session's kp_mex.  Ksysid.save_class (Ksysid.m:436-448) saves the whole object -
    % owners and the lift closures that captured them included; without Transient a loaded owner would carry the old uint64
    % and its delete() would hand kp_mex a pointer that was freed long ago, or never belonged to this process.  A loaded
    % owner is empty and releases nothing; KsysidHip.loadobj / KmpcHip.loadobj rebuild the device objects from their
    % descriptors.  (kp_mex also keeps a registry of live handles and rejects unknown ones with an error.)
    properties ( SetAccess = private , Transient )
        value;      % uint64 handle of kp_mex
        command;    % the kp_mex command that releases it
        parent;     % KpOwner that must outlive this one (or [])
    end
    methods
        function obj = KpOwner( value , command , parent )
            obj.value = value;
            obj.command = command;
            if nargin > 2
                obj.parent = parent;
            end
        end
        function release( obj )
            % early, explicit release (delete does the same when the object goes away)
            if ~isempty( obj.value )
                v = obj.value;
                obj.value = [];
                try
                    kp_mex( obj.command , v );
                catch
                    % the MEX file was cleared first: mexAtExit has released everything already
                end
            end
        end
        function delete( obj )
            obj.release();
        end
    end
    methods ( Static )
        function obj = loadobj( s )
            % whatever was saved (an object of an older version of this class arrives as a struct): an EMPTY owner
            obj = KpOwner( [] , '' );
            if isa( s , 'KpOwner' )
                obj = s;        % Transient properties are empty already
            end
        end
    end
end
